"""The balance QP (compute_grf's stance branch, S/A1RobotControl.cpp:377-444) under random COMBINATIONS of its constants (Q weights, R, mu, force limits) and of the OSQP
settings, against the oracle (GPU).  usage: soak_settings_balance.py [first_seed [count [qps]]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 500; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 40; n = int(sys.argv[3]) if len(sys.argv) > 3 else 192
tot = 0; bad = 0; worst = 0.0
for seed in range(lo, lo + cnt):
    rng = np.random.default_rng(seed)
    over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])), rho=float(10 ** rng.uniform(-2, 0.3)),
                sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])), adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])), adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
    over["eps_rel"] = over["eps_abs"]
    sc = pkg.scenarios.balance_random(n, seed=9000 + seed)
    qpo = orc.default_qp_params(); qpg = pkg.engine.BalanceConfig(); pkg.engine.load_library().a1mpc_default_balance_config(qpg)
    Q = [float(x) for x in np.array([1, 1, 1, 400, 400, 100]) * 10 ** rng.uniform(-0.5, 0.5, 6)]
    R = float(10 ** rng.uniform(-4, -2)); mu = float(rng.choice([0.7, 0.7, 0.4, 0.25])); fmin = float(rng.choice([0.0, 0.0, 5.0])); fmax = float(rng.choice([180.0, 120.0, 60.0]))
    for q_ in (qpo, qpg):
        (q_.Qw if hasattr(q_, "Qw") else q_.Q)[:] = Q; q_.R = R; q_.mu = mu; q_.F_min = fmin; q_.F_max = fmax
    cfg = pkg.make_config(pkg.scenarios.PARAM_SETS["gazebo"] | pkg.scenarios.MPC_CONSTANTS, 10, **over)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.balance_solve(sc["root_acc"], sc["R"], sc["Rz"], sc["foot"], sc["contact"], qp=qpg)
    st = orc.default_settings(**over); w = 0.0; mis = 0; its = 0
    for b in range(n):
        r = orc.balance_solve(qpo, st, sc["root_acc"][b], sc["R"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
        ok = out["iters"][b] == r["info"].iters and out["status"][b] == r["info"].status
        mis += int(not ok); its += r["info"].iters
        if ok: w = max(w, float(np.abs(out["grf"][b] - r["grf"]).max()))
    tot += n; bad += mis; worst = max(worst, w)
    print(seed, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in over.items()}, "R %.1e mu %.2f F [%.0f, %.0f]" % (R, mu, fmin, fmax),
          "| worst %.2e N, mismatching %d of %d, mean iters %.1f" % (w, mis, n, its / n), flush=True)
print("TOTAL", tot, "balance QPs in", cnt, "random combinations: worst %.3e N over QPs with equal iteration count and status, mismatching: %d" % (worst, bad))
