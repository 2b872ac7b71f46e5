#!/usr/bin/env python3
"""The accuracy yardstick of extended_check.py applied to a combination of soak_settings.py (by seed): for the QPs of that combination on which the GPU engine and the
double-precision oracle differ most, who is closer to the same OSQP iterate sequence computed in x87 extended precision?  GPU box.  usage: extended_check_settings.py seed [seed ...]"""
import ctypes as C, json, os, subprocess, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
HERE = os.path.dirname(os.path.abspath(orc.__file__))
X87 = os.path.join(HERE, "liba1mpc_oracle_x87.so")
if not os.path.exists(X87):
    subprocess.check_call(["make", "-C", HERE, "liba1mpc_oracle_x87.so"], stdout=subprocess.DEVNULL)
L = C.CDLL(X87); LD = C.c_longdouble
class SettingsX(C.Structure):
    _fields_ = [(k, LD) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf", "adaptive_rho_tolerance")] + \
               [(k, C.c_int32) for k in ("max_iter", "scaling", "check_termination", "adaptive_rho", "adaptive_rho_interval", "warm_start", "linsys", "reserved_")]
class InfoX(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("iters", "status", "rho_updates", "nfact")] + [(k, LD) for k in ("pri_res", "dua_res", "rho_final")] + [("reinit", C.c_int32), ("pad_", C.c_int32)]
class MpcParamsX(C.Structure):
    _fields_ = [("horizon", C.c_int32), ("dt", LD), ("mu", LD), ("fz_min", LD), ("fz_max", LD), ("q", LD * 13), ("r", LD * 12), ("mass", LD), ("inertia", LD * 9)]
ldp = lambda a: a.ctypes.data_as(C.POINTER(LD))
def solve_x87(p, h, over, x0, xref, R, foot, contact):
    pr = MpcParamsX(); pr.horizon = h; pr.dt = p["dt"]; pr.mu = p["mu"]; pr.fz_min = p["fz_min"]; pr.fz_max = p["fz_max"]; pr.mass = p["mass"]
    for i, v in enumerate(p["q"]): pr.q[i] = v
    for i, v in enumerate(p["r"]): pr.r[i] = v
    for i, v in enumerate(np.asarray(p["inertia"]).reshape(9)): pr.inertia[i] = v
    st = SettingsX(); L.orc_default_settings(C.byref(st))
    for k, v in over.items(): setattr(st, k, v)
    a = lambda v: np.ascontiguousarray(np.asarray(v, dtype=np.float64).astype(np.longdouble))
    X0, XR, RW, FT = a(x0), a(xref), a(R), a(foot); ct = np.ascontiguousarray(contact, dtype=np.uint8)
    grf = np.zeros(12, np.longdouble); info = InfoX()
    L.orc_mpc_solve(C.byref(pr), C.byref(st), ldp(X0), ldp(XR), ldp(RW), ldp(FT), C.c_int(0), ct.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(0), ldp(grf), None, None, None, None, C.byref(info))
    return grf.astype(np.float64), info.iters
rows = []; n = 256
for seed in [int(s) for s in sys.argv[1:]] or [1568]:
    rng = np.random.default_rng(seed)
    H = int(rng.choice([10, 10, 16, 20]))
    over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])), rho=float(10 ** rng.uniform(-2, 0.3)),
                sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])), adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])), adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4, 1e-5])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
    over["eps_rel"] = over["eps_abs"]
    gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[H]
    sc = gen(nb=n, seed=7000 + seed)
    p = dict(sc["params"], mu=float(rng.choice([0.3, 0.3, 0.6, 0.15])), fz_min=float(rng.choice([0.0, 0.0, 0.0, 5.0])), fz_max=float(rng.choice([180.0, 180.0, 120.0, 60.0])))
    with pkg.Engine(pkg.make_config(p, H, warm_start=0, **over), n, 0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    pr = orc.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = orc.mpc_solve_batch(pr, orc.default_settings(**over), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    gg = out["grf"].reshape(n, 12); oo = ref["grf"].reshape(n, 12); dd = np.abs(gg - oo).max(1)
    for i in np.argsort(-dd)[:3]:
        x, it = solve_x87(p, H, over, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i])
        rows.append({"seed": seed, "h": H, "qp": int(i), "iters_x87": int(it), "iters_gpu": int(out["iters"].ravel()[i]), "iters_f64_oracle": int(ref["iters"].ravel()[i]),
                     "gpu_vs_oracle_N": float(dd[i]), "gpu_vs_x87_N": float(np.abs(gg[i] - x).max()), "oracle_vs_x87_N": float(np.abs(oo[i] - x).max())})
print(json.dumps({"note": "soak_settings.py combinations above the parity bar: the three QPs with the largest GPU-vs-oracle difference, re-solved in x87 extended precision", "rows": rows}))
