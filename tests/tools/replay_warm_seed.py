"""diagnostic (GPU): replays one sequence of soak_settings_warm.py (seed, tick) with the ORACLE's carried (x, y, rho) injected into the engine before that tick -- separates a deviation of the
solve itself from the amplification of last-bit differences in the carried state (seed 318, tick 23: 3e-3 N in the free-running sequence, 1e-12 N with the same warm state on both sides;
the oracle's own two linear-system back ends differ by 1.7e-6 N on that solve).  usage: replay_warm_seed.py [seed [tick]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 318; TSTOP = int(sys.argv[2]) if len(sys.argv) > 2 else 23
robots, ticks = 48, 24
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
st = orc.default_settings(warm_start=1, **over); st1 = orc.default_settings(warm_start=1, linsys=1, **over)
ws = [(np.zeros(12 * H), np.zeros(20 * H), None) for _ in range(robots)]
print(over)
with pkg.Engine(pkg.make_config(p, H, warm_start=1, **over), robots, 0) as eng:
    for t in range(ticks):
        if t > 0:
            sc["x0"][:, :12] += rng.normal(0, 1.5e-3, (robots, 12)); sc["foot"] += rng.normal(0, 5e-4, (robots, 12))
        if t % 9 == 5:
            flip = rng.random(robots) < 0.5
            c = sc["contact"].copy(); c[flip] = 1 - c[flip]; c[c.sum(1) == 0] = [1, 0, 0, 1]; sc["contact"] = c
        if t % 11 == 7:
            j = rng.random(robots) < 0.15
            sc["x0"][j, 6:12] += rng.normal(0, 0.3, (int(j.sum()), 6))
        if t == TSTOP:   # same warm state on both sides
            eng.set_warm_start(np.array([w[0] for w in ws]), np.array([w[1] for w in ws]), np.array([w[2] if w[2] is not None else 0.0 for w in ws]))
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        for b in range(robots):
            wx, wy, rho = ws[b]
            o = orc.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], warm_x=wx, warm_y=wy, warm_rho=rho)
            if t == TSTOP:
                o1 = orc.mpc_solve(pr, st1, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], warm_x=wx, warm_y=wy, warm_rho=rho)
                d = float(np.abs(out["grf"][b] - o["grf"]).max()); d1 = float(np.abs(o1["grf"] - o["grf"]).max()); du = float(np.abs(out["u"][b] - o["u"]).max())
                if d > 1e-7 or d1 > 1e-7:
                    print("tick", t, "robot", b, "gpu-oracle %.2e (u %.2e)" % (d, du), "oracle1-oracle0 %.2e" % d1, "iters gpu", out["iters"][b], "orc", o["info"].iters, o1["info"].iters,
                          "rho_final", o["info"].rho_final, o1["info"].rho_final, "rho_updates", o["info"].rho_updates, "status", o["info"].status, "contact", sc["contact"][b], "warm rho", rho)
            ws[b] = (o["warm_x"], o["warm_y"], o["rho"])
    x, y, rho_g = eng.get_warm_start(robots)
    print("rho carried gpu vs oracle, max rel diff:", max(abs(rho_g[b] - ws[b][2]) / ws[b][2] for b in range(robots)))
