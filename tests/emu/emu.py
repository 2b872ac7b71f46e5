"""ctypes front-end of tests/emu/liba1mpc_emu.so -- TEST INFRASTRUCTURE (CPU execution of the solver source
on host fibers; see emu_harness.cpp).  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "a1-qp-mpc-controller_amd", "csrc")
_LIB = os.path.join(_HERE, "liba1mpc_emu.so")


class DeviceParams(C.Structure):  # mirrors a1mpc::DeviceParams (csrc/a1mpc_solver.hpp)
    _fields_ = [("dt", C.c_double), ("mu", C.c_double), ("fz_min", C.c_double), ("fz_max", C.c_double),
                ("q2", C.c_double * 12), ("r2", C.c_double * 12), ("mass", C.c_double), ("inertia", C.c_double * 9),
                ("rho0", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("eps_abs", C.c_double),
                ("eps_rel", C.c_double), ("adaptive_rho_tol", C.c_double), ("max_iter", C.c_int32),
                ("check_every", C.c_int32), ("adaptive_rho", C.c_int32), ("adaptive_rho_every", C.c_int32),
                ("scaling_iters", C.c_int32), ("warm_start", C.c_int32)]


KNOWN_VARIANTS = {"fullsweep": ["-DA1X_FULL_SWEEP"], "poison": ["-DA1X_POISON"]}   # the extra builds tests/test_emu_parity.py asks for (variant())


def _sources():
    return [os.path.join(_HERE, "emu_harness.cpp"), os.path.join(_HERE, "a1mpc_rowops.hpp"), os.path.join(_CSRC, "a1mpc_solver.hpp"), os.path.join(_CSRC, "a1mpc_tables.hpp")]


def _stale(path):
    return not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in _sources())


def _variant_path(tag):
    return os.path.join(_HERE, f"liba1mpc_emu_{tag}.so")


def _build_all(force=False, extra=None):
    """Compiles every stale library of the test double -- the default build and the known variants -- IN PARALLEL (one g++ each: ~4 minutes of template instantiation per build,
    three builds in a row were 11 minutes of the CPU suite) under one file lock: pytest-xdist workers that arrive later wait and find everything fresh.  A build is written to a
    temporary name and renamed, so a reader never maps a half-written library."""
    import fcntl
    jobs = {}
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        want = {_LIB: os.environ.get("A1_EMU_FLAGS", "").split()}
        want.update({_variant_path(t): list(f) for t, f in KNOWN_VARIANTS.items()})
        want.update(extra or {})
        for path, flags in want.items():
            if force or _stale(path):
                tmp = f"{path}.tmp{os.getpid()}"
                jobs[path] = (tmp, subprocess.Popen(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", _HERE, "-I", _CSRC, _sources()[0], "-o", tmp] + flags))
        for path, (tmp, proc) in jobs.items():
            if proc.wait() != 0:
                raise RuntimeError(f"g++ failed on the CPU test double ({os.path.basename(path)})")
            os.replace(tmp, path)
    return list(jobs)


def build(force=False):
    _build_all(force)
    return _LIB


def variant(flags, tag):
    """a second build of the test double with extra compiler flags (e.g. -DA1X_FULL_SWEEP), as its own library; use it with `using()`"""
    path = _variant_path(tag)
    _build_all(extra=None if KNOWN_VARIANTS.get(tag) == list(flags) else {path: list(flags)})
    v = C.CDLL(path)
    assert v.a1mpc_emu_sizeof_params() == C.sizeof(DeviceParams)
    return v


class using:
    """with emu.using(emu.variant(...)): the solve functions of this module run the given build"""
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        global _lib
        lib()
        self.old, _lib = _lib, self.v
        return self.v

    def __exit__(self, *a):
        global _lib
        _lib = self.old


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.a1mpc_emu_sizeof_params() == C.sizeof(DeviceParams)
    return _lib


def make_params(params, settings=None, **over):
    """params: scenario['params'] dict; settings: dict of OSQP settings (defaults = OSQP 0.6 defaults)."""
    st = dict(rho0=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_tol=5.0, max_iter=4000,
              check_every=25, adaptive_rho=1, adaptive_rho_every=25, scaling_iters=10, warm_start=0)
    st.update(settings or {})
    st.update(over)
    p = DeviceParams()
    p.dt, p.mu, p.fz_min, p.fz_max, p.mass = params["dt"], params["mu"], params["fz_min"], params["fz_max"], params["mass"]
    p.q2[:] = [2.0 * v for v in params["q"][:12]]
    p.r2[:] = [2.0 * v for v in params["r"]]
    p.inertia[:] = list(np.asarray(params["inertia"], dtype=float).reshape(9))
    for k, v in st.items():
        setattr(p, k, v)
    return p


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def carry_buffer(horizon, n):
    """zeroed Carry<H> records of n problems (the update path, warm_start = 2)"""
    return np.zeros((n, int(lib().a1mpc_emu_carry_stride(int(horizon)))))


def solve(sc, n=None, settings=None, warm=None, split_rows=0, twin=False, contact_schedule=None, carry=None, quad=False, latency=False, **over):
    """split_rows > 0: run the two-kernel pipeline (set-up kernel, then `split_rows` persistent ADMM rows) instead of the fused path;
    twin: the iterations run on main / twin PAIRS of rows (RowSolver<.., TWIN>: what the device kernels do for H > 1);
    quad: ... on a QUAD of rows, 64 fibers in the device's lane order (RowSolver<.., QUAD>: the device kernels with one QP per wavefront, H = 16 / 20; the set-up of the
    fused path is shared by the four rows, like the latency kernel's);
    latency: the latency kernel of batches <= 256 QPs -- the four rows share the set-up, then rows 0 / 2 solve as a pair (rows 1 / 3 retire) or, at H = 16 / 20, all four as a quad;
    contact_schedule: (n, 4h) per-step contacts on the FAST path (feet step-invariant) instead of sc["contact"]"""
    h = sc["horizon"]
    n = len(sc["x0"]) if n is None else n
    P = make_params(sc["params"], settings, **over)
    grf = np.zeros((n, 12)); u = np.zeros((n, 12 * h))
    iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32); nfact = np.zeros(n, np.int32)
    wx = wy = rho = None
    if warm is not None:
        wx, wy, rho = warm
    contact = sc["contact"]
    if contact_schedule is not None:
        contact = np.ascontiguousarray(contact_schedule, dtype=np.uint8).reshape(-1, 4 * h)
    lib().a1mpc_emu_set_contact_stride(4 if contact_schedule is not None else 0)
    lib().a1mpc_emu_set_carry.argtypes = [C.c_void_p]
    lib().a1mpc_emu_set_carry(None if carry is None else carry.ctypes.data)   # update path (warm_start = 2): carried in place
    try:
        return _solve(sc, n, h, P, grf, u, iters, status, nfact, wx, wy, rho, contact, split_rows, twin or quad, quad, latency)
    finally:
        lib().a1mpc_emu_set_contact_stride(0); lib().a1mpc_emu_set_carry(None)


def _solve(sc, n, h, P, grf, u, iters, status, nfact, wx, wy, rho, contact, split_rows, twin, quad=False, latency=False):
    if split_rows:
        lib().a1mpc_emu_set_twin(2 if quad else 0)
        rc = lib().a1mpc_emu_solve_split(C.byref(P), h, n, -int(split_rows) if twin else int(split_rows), _p(sc["x0"]), _p(sc["xref"]), _p(sc["R"]), _p(sc["foot"]),
                                         _p(contact, C.c_uint8), _p(grf), _p(u), _p(wx), _p(wy), _p(rho), _p(iters, C.c_int32),
                                         _p(status, C.c_int32), _p(nfact, C.c_int32))
        lib().a1mpc_emu_set_twin(0)
        assert rc == 0
        return dict(grf=grf, u=u, iters=iters, status=status, nfact=nfact)
    lib().a1mpc_emu_set_twin(3 if latency else (2 if quad else (1 if twin else 0)))
    try:
        rc = lib().a1mpc_emu_solve(C.byref(P), h, n, _p(sc["x0"]), _p(sc["xref"]), _p(sc["R"]), _p(sc["foot"]),
                                   _p(contact, C.c_uint8), _p(grf), _p(u), _p(wx), _p(wy), _p(rho), _p(iters, C.c_int32),
                                   _p(status, C.c_int32), _p(nfact, C.c_int32))
    finally:
        lib().a1mpc_emu_set_twin(0)
    assert rc == 0
    return dict(grf=grf, u=u, iters=iters, status=status, nfact=nfact)


def solve_gen(sc, foot, foot_stride, contact, contact_stride, n=None, settings=None, warm=None, twin=False, quad=False, carry=None, latency=False, **over):
    """the general path: foot (n, 12) or (n, 12h) with foot_stride 0 / 12; contact (n, 4) or (n, 4h) with contact_stride 0 / 4;
    carry: Carry<H> records (carry_buffer) for the update path, warm_start = 2 (twin / quad runs)"""
    h = sc["horizon"]
    n = len(sc["x0"]) if n is None else n
    P = make_params(sc["params"], settings, **over)
    foot = np.ascontiguousarray(foot, dtype=np.float64); contact = np.ascontiguousarray(contact, dtype=np.uint8)
    grf = np.zeros((n, 12)); u = np.zeros((n, 12 * h))
    iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32); nfact = np.zeros(n, np.int32)
    wx = wy = rho = None
    if warm is not None:
        wx, wy, rho = warm
    lib().a1mpc_emu_set_twin(3 if latency else (2 if quad else (1 if twin else 0)))   # latency: the general path's latency kernel (h = 10)
    lib().a1mpc_emu_set_carry.argtypes = [C.c_void_p]
    lib().a1mpc_emu_set_carry(None if carry is None else carry.ctypes.data)
    rc = lib().a1mpc_emu_solve_gen(C.byref(P), h, n, _p(sc["x0"]), _p(sc["xref"]), _p(sc["R"]), _p(foot), int(foot_stride), _p(contact, C.c_uint8),
                                   int(contact_stride), _p(grf), _p(u), _p(wx), _p(wy), _p(rho), _p(iters, C.c_int32), _p(status, C.c_int32),
                                   _p(nfact, C.c_int32))
    lib().a1mpc_emu_set_twin(0); lib().a1mpc_emu_set_carry(None)
    assert rc == 0
    return dict(grf=grf, u=u, iters=iters, status=status, nfact=nfact)


def solve_gen_split(sc, foot, foot_stride, contact, contact_stride, rows=2, n=None, settings=None, quad=False, **over):
    """the general path's split pipeline: its own set-up kernel, then `rows` persistent main / twin pairs draining the queue"""
    h = sc["horizon"]
    n = len(sc["x0"]) if n is None else n
    P = make_params(sc["params"], settings, **over)
    foot = np.ascontiguousarray(foot, dtype=np.float64); contact = np.ascontiguousarray(contact, dtype=np.uint8)
    grf = np.zeros((n, 12)); u = np.zeros((n, 12 * h))
    iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32); nfact = np.zeros(n, np.int32)
    lib().a1mpc_emu_set_twin(2 if quad else 0)
    rc = lib().a1mpc_emu_solve_gen_split(C.byref(P), h, n, int(rows), _p(sc["x0"]), _p(sc["xref"]), _p(sc["R"]), _p(foot), int(foot_stride), _p(contact, C.c_uint8),
                                         int(contact_stride), _p(grf), _p(u), _p(iters, C.c_int32), _p(status, C.c_int32), _p(nfact, C.c_int32))
    lib().a1mpc_emu_set_twin(0)
    assert rc == 0
    return dict(grf=grf, u=u, iters=iters, status=status, nfact=nfact)


def balance_params(qp=None, settings=None, **over):
    """DeviceParams of the balance QP (S/A1RobotControl.cpp:11-15): the H = 1 member of the family with
    dt = 0, wrench weights (torque first) in q2[6:12], R in r2."""
    qp = qp or dict(Q=[1.0, 1.0, 1.0, 400.0, 400.0, 100.0], R=1e-3, mu=0.7, F_min=0.0, F_max=180.0)
    prm = dict(dt=0.0, mu=qp["mu"], fz_min=qp["F_min"], fz_max=qp["F_max"], mass=1.0, inertia=np.eye(3).reshape(9),
               q=[0.0] * 12, r=[0.0] * 12)
    p = make_params(prm, settings, **over)
    p.q2[:] = [0.0] * 6 + list(qp["Q"][3:6]) + list(qp["Q"][0:3])
    p.r2[:] = [qp["R"]] * 12
    p.warm_start = 0
    return p


def balance_solve(sc, n=None, settings=None, **over):
    n = len(sc["root_acc"]) if n is None else n
    P = balance_params(None, settings, **over)
    grf = np.zeros((n, 12)); f = np.zeros((n, 12)); iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
    rc = lib().a1mpc_emu_balance(C.byref(P), n, _p(sc["root_acc"]), _p(sc["R"]), _p(sc["Rz"]), _p(sc["foot"]),
                                 _p(sc["contact"], C.c_uint8), _p(grf), _p(f), _p(iters, C.c_int32), _p(status, C.c_int32))
    assert rc == 0
    return dict(grf=grf, f_world=f, iters=iters, status=status)


def solve_ticks(sc, n=None, settings=None, **over):
    """N1: compact tick records in, x0 / x_ref built by the solver source (horizon 10)."""
    n = len(sc["tick"]) if n is None else n
    P = make_params(sc["params"], settings, **over)
    grf = np.zeros((n, 12)); u = np.zeros((n, 120)); iters = np.zeros(n, np.int32); status = np.zeros(n, np.int32)
    rc = lib().a1mpc_emu_solve_ticks(C.byref(P), 10, n, _p(sc["tick"]), _p(sc["R"]), _p(sc["foot"]), _p(sc["contact"], C.c_uint8), _p(grf), _p(u),
                                     _p(iters, C.c_int32), _p(status, C.c_int32))
    assert rc == 0
    return dict(grf=grf, u=u, iters=iters, status=status)
