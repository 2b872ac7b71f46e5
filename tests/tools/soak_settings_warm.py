"""Warm-started tick sequences under random COMBINATIONS of OSQP settings, in both warm-start semantics (1: fresh set-up + warm start; 2: the reference's update path),
robot by robot against the oracle (GPU).  Slowly drifting states with contact flips and pushes, like soak_update_path.py.
usage: soak_settings_warm.py [first_seed [count [robots [ticks]]]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 300; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 24
robots = int(sys.argv[3]) if len(sys.argv) > 3 else 48; ticks = int(sys.argv[4]) if len(sys.argv) > 4 else 24
tot = 0; bad = 0; worst = 0.0
for seed in range(lo, lo + cnt):
    rng = np.random.default_rng(seed)
    mode = 1 + seed % 2
    H = 10 if mode == 2 else int(rng.choice([10, 10, 16, 20]))
    over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])), rho=float(10 ** rng.uniform(-2, 0.3)),
                sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])), adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])), adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
    over["eps_rel"] = over["eps_abs"]
    gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[H]
    sc = gen(nb=robots, seed=8000 + seed); p = sc["params"]
    pr = orc.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = orc.default_settings(warm_start=1, **over)
    carries = [orc.update_carry(H) for _ in range(robots)]
    ws = [(np.zeros(12 * H), np.zeros(20 * H), None) for _ in range(robots)]
    w = 0.0; mis = 0; its = 0; per_tick = []
    with pkg.Engine(pkg.make_config(p, H, warm_start=mode, **over), robots, 0) as eng:
        for t in range(ticks):
            if t > 0:
                sc["x0"][:, :12] += rng.normal(0, 1.5e-3, (robots, 12)); sc["foot"] += rng.normal(0, 5e-4, (robots, 12))
            if t % 9 == 5:
                flip = rng.random(robots) < 0.5
                c = sc["contact"].copy(); c[flip] = 1 - c[flip]; c[c.sum(1) == 0] = [1, 0, 0, 1]; sc["contact"] = c
            if t % 11 == 7:
                j = rng.random(robots) < 0.15
                sc["x0"][j, 6:12] += rng.normal(0, 0.3, (int(j.sum()), 6))
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); w0 = w; w = 0.0
            for b in range(robots):
                if mode == 2:
                    o = orc.mpc_solve_update(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], carries[b])
                else:
                    wx, wy, rho = ws[b]
                    o = orc.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], warm_x=wx, warm_y=wy, warm_rho=rho)
                    ws[b] = (o["warm_x"], o["warm_y"], o["rho"])
                ok = out["iters"][b] == o["info"].iters and out["status"][b] == o["info"].status
                mis += int(not ok); its += o["info"].iters
                if ok: w = max(w, float(np.abs(out["grf"][b] - o["grf"]).max()))
            per_tick.append(w); w = max(w, w0)
    tot += robots * ticks; bad += mis; worst = max(worst, w)
    print(seed, "warm_start", mode, "h", H, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in over.items()},
          "| worst %.2e N, mismatching %d of %d, mean iters %.1f" % (w, mis, robots * ticks, its / (robots * ticks)),
          ("per tick: " + " ".join("%.0e" % x for x in per_tick)) if os.environ.get("A1_SOAK_PER_TICK") else "", flush=True)
print("TOTAL", tot, "warm-started solves in", cnt, "random setting combinations: worst %.3e N over solves with equal iteration count and status, mismatching: %d" % (worst, bad))
