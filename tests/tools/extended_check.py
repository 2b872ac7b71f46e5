#!/usr/bin/env python3
"""Accuracy yardstick: for the QPs where the GPU engine and the double-precision oracle differ most, who is closer to the same OSQP iterate
sequence computed in x87 extended precision (oracle source built with -DORC_EXTENDED)?  Runs on the GPU box.  Prints one JSON line.
TEST / ANALYSIS INFRASTRUCTURE: loads oracle/ like the tests do."""
import ctypes as C, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
HERE = os.path.dirname(os.path.abspath(orc.__file__))
X87 = os.path.join(HERE, "liba1mpc_oracle_x87.so")
if not os.path.exists(X87):
    subprocess.check_call(["make", "-C", HERE, "liba1mpc_oracle_x87.so"], stdout=subprocess.DEVNULL)
L = C.CDLL(X87)
LD = C.c_longdouble


class SettingsX(C.Structure):
    _fields_ = [(k, LD) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf", "adaptive_rho_tolerance")] + \
               [(k, C.c_int32) for k in ("max_iter", "scaling", "check_termination", "adaptive_rho", "adaptive_rho_interval", "warm_start")]


class InfoX(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("iters", "status", "rho_updates", "nfact")] + [(k, LD) for k in ("pri_res", "dua_res", "rho_final")]


class MpcParamsX(C.Structure):
    _fields_ = [("horizon", C.c_int32), ("dt", LD), ("mu", LD), ("fz_min", LD), ("fz_max", LD), ("q", LD * 13), ("r", LD * 12), ("mass", LD), ("inertia", LD * 9)]


def ldp(a):
    return a.ctypes.data_as(C.POINTER(LD))


def solve_x87(p, h, x0, xref, R, foot, contact):
    pr = MpcParamsX(); pr.horizon = h; pr.dt = p["dt"]; pr.mu = p["mu"]; pr.fz_min = p["fz_min"]; pr.fz_max = p["fz_max"]; pr.mass = p["mass"]
    for i, v in enumerate(p["q"]): pr.q[i] = v
    for i, v in enumerate(p["r"]): pr.r[i] = v
    for i, v in enumerate(np.asarray(p["inertia"]).reshape(9)): pr.inertia[i] = v
    st = SettingsX(); L.orc_default_settings(C.byref(st))
    a = lambda v: np.ascontiguousarray(np.asarray(v, dtype=np.float64).astype(np.longdouble))
    X0, XR, RW, FT = a(x0), a(xref), a(R), a(foot)
    ct = np.ascontiguousarray(contact, dtype=np.uint8)
    grf = np.zeros(12, np.longdouble); info = InfoX()
    L.orc_mpc_solve(C.byref(pr), C.byref(st), ldp(X0), ldp(XR), ldp(RW), ldp(FT), C.c_int(0), ct.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(0),
                    ldp(grf), None, None, None, None, C.byref(info))
    return grf.astype(np.float64), info.iters, info.nfact


rows = []
for seed in (0xA1 + 3, 1002, 1003, 1007):
    n = 4096
    sc = pkg.scenarios.config3_random_flat(nb=n, seed=seed); p = sc["params"]
    cfg = pkg.make_config(p, 10, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    pr = orc.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = orc.mpc_solve_batch(pr, orc.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    gg = out["grf"].reshape(n, 12); oo = ref["grf"].reshape(n, 12)
    dd = np.abs(gg - oo).max(1)
    for i in np.argsort(-dd)[:3]:
        x, it, nf = solve_x87(p, 10, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i])
        rows.append({"seed": int(seed), "qp": int(i), "iters_x87": int(it), "iters_gpu": int(out["iters"].ravel()[i]), "iters_f64_oracle": int(ref["iters"].ravel()[i]),
                     "gpu_vs_oracle_N": float(dd[i]), "gpu_vs_x87_N": float(np.abs(gg[i] - x).max()), "oracle_vs_x87_N": float(np.abs(oo[i] - x).max()),
                     "max_force_N": float(np.abs(x).max())})
print(json.dumps({"note": "largest GPU-vs-oracle differences of 4 x 4096 random QPs (config3 generator), each re-solved in x87 extended precision",
                  "rows": rows}))
