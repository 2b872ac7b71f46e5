"""Random COMBINATIONS of OSQP settings, friction / force limits, weight sets and horizons against the oracle (GPU): the single-setting cases of
tests/test_gpu_settings.py::test_non_default_osqp_settings draw one knob at a time.  usage: soak_settings.py [first_seed [count [qps_per_case]]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 100; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 60; n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
worst = 0.0; bad = 0; tot = 0; unsolved = 0
HS = [int(v) for v in os.environ["A1_SOAK_HORIZONS"].split(",")] if os.environ.get("A1_SOAK_HORIZONS") else [10, 10, 16, 20]   # (A1_SOAK_HORIZONS=4,6,8,12,14: the extended horizons)
for seed in range(lo, lo + cnt):
    rng = np.random.default_rng(seed)
    H = int(rng.choice(HS))
    over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])), rho=float(10 ** rng.uniform(-2, 0.3)),
                sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])), adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])), adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4, 1e-5])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
    over["eps_rel"] = over["eps_abs"]
    gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}.get(H)
    sc = gen(nb=n, seed=7000 + seed) if gen else (pkg.scenarios.config5_divergent if seed % 2 else pkg.scenarios.config3_random_flat)(nb=n, seed=7000 + seed, horizon=H)
    p = dict(sc["params"], mu=float(rng.choice([0.3, 0.3, 0.6, 0.15])), fz_min=float(rng.choice([0.0, 0.0, 0.0, 5.0])), fz_max=float(rng.choice([180.0, 180.0, 120.0, 60.0])))
    cfg = pkg.make_config(p, H, warm_start=0, **over)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    pr = orc.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = orc.mpc_solve_batch(pr, orc.default_settings(**over), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    dd = np.abs(out["grf"].reshape(n, 12) - ref["grf"].reshape(n, 12)).max(1)
    same = out["iters"].ravel() == ref["iters"].ravel(); steq = out["status"].ravel() == ref["status"].ravel()
    ok = same & steq
    worst = max(worst, float(dd[ok].max()) if ok.any() else 0.0); bad += int((~ok).sum()); tot += n; unsolved += int((ref["status"] != 1).sum())
    print(seed, "h", H, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in over.items()}, "mu %.2f fz [%.0f, %.0f]" % (p["mu"], p["fz_min"], p["fz_max"]),
          "| max %.2e  same iters %.4f  status eq %.4f  mean iters %.1f" % (dd.max(), same.mean(), steq.mean(), ref["iters"].mean()), flush=True)
print("TOTAL", tot, "QPs in", cnt, "random setting combinations: worst %.3e N over QPs with equal iteration count and status, mismatching: %d, not 'solved' in the oracle: %d" % (worst, bad, unsolved))
