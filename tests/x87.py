"""The oracle source compiled in x87 extended precision (oracle/liba1mpc_oracle_x87.so: every double widened to the 80-bit long double, `make -C oracle
liba1mpc_oracle_x87.so`) -- TEST INFRASTRUCTURE, an accuracy yardstick: the same OSQP iterate sequence with 2048x less rounding error, against which the
double-precision oracle and the GPU engine can both be measured where THEY disagree.  Never the pass / fail checker of a result by itself."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_PATH = os.path.join(_HERE, "liba1mpc_oracle_x87.so")
LD = C.c_longdouble


class SettingsX(C.Structure):   # orc_settings with double -> long double
    _fields_ = [(k, LD) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf", "adaptive_rho_tolerance")] + \
               [(k, C.c_int32) for k in ("max_iter", "scaling", "check_termination", "adaptive_rho", "adaptive_rho_interval", "warm_start", "linsys", "reserved_")]


class InfoX(C.Structure):       # orc_info
    _fields_ = [(k, C.c_int32) for k in ("iters", "status", "rho_updates", "nfact")] + [(k, LD) for k in ("pri_res", "dua_res", "rho_final")] + \
               [("reinit", C.c_int32), ("pad_", C.c_int32)]


class MpcParamsX(C.Structure):  # orc_mpc_params
    _fields_ = [("horizon", C.c_int32), ("dt", LD), ("mu", LD), ("fz_min", LD), ("fz_max", LD), ("q", LD * 13), ("r", LD * 12), ("mass", LD), ("inertia", LD * 9)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "a1mpc_oracle.c")
        if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liba1mpc_oracle_x87.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_PATH)
    return _lib


def _ld(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.longdouble))


def _ldp(a):
    return a.ctypes.data_as(C.POINTER(LD))


def params(p, h):
    pr = MpcParamsX(); pr.horizon = int(h); pr.dt = p["dt"]; pr.mu = p["mu"]; pr.fz_min = p["fz_min"]; pr.fz_max = p["fz_max"]; pr.mass = p["mass"]
    for i, v in enumerate(p["q"]): pr.q[i] = v
    for i, v in enumerate(p["r"]): pr.r[i] = v
    for i, v in enumerate(np.asarray(p["inertia"], dtype=float).reshape(9)): pr.inertia[i] = v
    return pr


def settings(**over):
    st = SettingsX(); lib().orc_default_settings(C.byref(st))
    for k, v in over.items():
        setattr(st, k, v)
    return st


def mpc_solve_update(pr, st, x0, xref, Rw, foot, contact, carry):
    """one tick on the update path from the double-precision workspace `carry` (oracle.update_carry layout; not modified): dict(grf, u, iters, status)"""
    h = pr.horizon
    c = _ld(carry)
    grf = np.zeros(12, np.longdouble); u = np.zeros(12 * h, np.longdouble); info = InfoX()
    ct = np.ascontiguousarray(contact, dtype=np.uint8)
    X0, XR, RW, FT = _ld(x0), _ld(xref), _ld(Rw), _ld(foot)
    lib().orc_mpc_solve_update(C.byref(pr), C.byref(st), _ldp(X0), _ldp(XR), _ldp(RW), _ldp(FT), ct.ctypes.data_as(C.POINTER(C.c_uint8)), _ldp(grf), _ldp(u), _ldp(c),
                               C.byref(info))
    return dict(grf=grf.astype(np.float64), u=u.astype(np.float64), iters=int(info.iters), status=int(info.status))


def mpc_solve(pr, st, x0, xref, Rw, foot, contact):
    """one cold solve (orc_mpc_solve in extended precision): dict(grf, iters, status)"""
    grf = np.zeros(12, np.longdouble); info = InfoX()
    ct = np.ascontiguousarray(contact, dtype=np.uint8)
    X0, XR, RW, FT = _ld(x0), _ld(xref), _ld(Rw), _ld(foot)
    lib().orc_mpc_solve(C.byref(pr), C.byref(st), _ldp(X0), _ldp(XR), _ldp(RW), _ldp(FT), C.c_int(0), ct.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(0), _ldp(grf),
                        None, None, None, None, C.byref(info))
    return dict(grf=grf.astype(np.float64), iters=int(info.iters), status=int(info.status))
