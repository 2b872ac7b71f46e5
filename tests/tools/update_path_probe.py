"""TEST INFRASTRUCTURE (uses oracle/): how far is the reference's UPDATE PATH from the "fresh set-up + osqp_warm_start" restatement the engine implements?
On every tick after the first the reference calls updateHessianMatrix / updateGradient / update*Bound on its persistent OsqpEigen solver
(S/A1RobotControl.cpp:533-538): OSQP re-equilibrates with the previous tick's gradient still in the workspace and keeps the carried iterates in the
previous tick's scaling (oracle: osqp_solve_impl with a carry).  The engine (and orc_mpc_solve) set every tick up from scratch and warm-start from the
previous tick's UNSCALED (x, y, rho).  Both are warm starts of the same QP; this script measures the difference of the returned forces and of the
iteration counts over warm-started trot sequences (configs[1]).  Usage: python tests/tools/update_path_probe.py [nticks] -> profiles/r02_update_path_deviation.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

pkg = g.load_package(); oracle = g.load_oracle()
nticks = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
H = 10


def smooth_sequence(n, seed=11):
    """the same trot as config2, but the state moves like a robot's: a slow random walk instead of independent noise per tick"""
    scen = pkg.scenarios
    seq = scen.config2_trot_sequence(n)
    rng = np.random.default_rng(seed)
    lp = lambda sig, k: np.cumsum(rng.normal(0, sig / 30.0, (n, k)), axis=0)  # random walk, ~sig after 900 ticks
    p = seq["params"]
    euler = np.clip(lp(0.02, 3), -0.1, 0.1); pos = np.zeros((n, 3)); pos[:, 2] = 0.3 + np.clip(lp(0.01, 1)[:, 0], -0.03, 0.03)
    ang = np.clip(lp(0.1, 3), -0.5, 0.5); lin = np.clip(lp(0.05, 3), -0.3, 0.3); lin[:, 0] += 0.3
    pos[:, 0] = np.cumsum(lin[:, 0]) * 0.0025
    R = scen.rot_zyx(euler[:, 0], euler[:, 1], euler[:, 2])
    vd = np.tile(np.array([[0.3, 0.0, 0.0]]), (n, 1)); zero = np.zeros((n, 3))
    seq = dict(seq)
    seq["x0"] = scen.pack_x0(euler, pos, ang, lin)
    seq["xref"] = scen.build_reference(H, scen.MPC_CONSTANTS["dt"], euler, pos, R, zero, vd, zero, np.full(n, 0.3))
    seq["R"] = R.reshape(n, 9)
    seq["foot"] = np.einsum("bij,lj->bli", R.reshape(n, 3, 3), np.array(p["foot"], dtype=float)).reshape(n, 12)
    return seq


def run(seq, n):
    p = seq["params"]
    pr = oracle.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    wx = np.zeros(12 * H); wy = np.zeros(20 * H); rho = 0.0
    carry = oracle.update_carry(H)
    d, it_a, it_b, fz = [], [], [], []
    for k in range(n):
        a = oracle.mpc_solve(pr, st, seq["x0"][k], seq["xref"][k], seq["R"][k], seq["foot"][k], seq["contact"][k], warm_x=wx, warm_y=wy, warm_rho=rho)
        wx, wy, rho = a["warm_x"], a["warm_y"], a["rho"]
        b = oracle.mpc_solve_update(pr, st, seq["x0"][k], seq["xref"][k], seq["R"][k], seq["foot"][k], seq["contact"][k], carry)
        d.append(float(np.abs(a["grf"] - b["grf"]).max())); it_a.append(a["info"].iters); it_b.append(b["info"].iters); fz.append(float(np.abs(a["grf"]).max()))
    d = np.array(d); it_a = np.array(it_a); it_b = np.array(it_b)
    return {"ticks": n, "first_tick_identical": bool(d[0] == 0.0 and it_a[0] == it_b[0]),
            "max_abs_dgrf_N": {"median": float(np.median(d[1:])), "p99": float(np.percentile(d[1:], 99)), "max": float(d[1:].max())},
            "largest_force_N_median": float(np.median(fz)),
            "ticks_with_different_iteration_count": int((it_a != it_b).sum()), "mean_iters_restatement": float(it_a.mean()), "mean_iters_update_path": float(it_b.mean())}


out = {"what": __doc__.split("Usage")[0].strip(),
       "config2_iid_noise_per_tick": run(pkg.scenarios.config2_trot_sequence(nticks), nticks),
       "config2_slow_random_walk": run(smooth_sequence(nticks), nticks),
       "yardstick": "OSQP's own slack at its default 1e-3 tolerances: the returned forces are ~1 N (median) away from the QP's optimum (DESIGN 5)"}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_update_path_deviation.json"), "w"), indent=1)
