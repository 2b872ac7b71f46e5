import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
for gen, h, n in (("config4_random_h16", 16, 1024), ("config5_divergent", 20, 1024), ("config3_random_flat", 10, 4096)):
    for seed in (11, 12, 13):
        sc = getattr(pkg.scenarios, gen)(nb=n, seed=seed); p = sc["params"]
        cfg = pkg.make_config(p, h, warm_start=0)
        with pkg.Engine(cfg, n, 0) as eng:
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        pr = orc.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
        ref = orc.mpc_solve_batch(pr, orc.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        dd = np.abs(out["grf"].reshape(n, 12) - ref["grf"].reshape(n, 12)).max(1)
        print(gen, "seed", seed, "max %.2e p99.9 %.2e median %.2e same iters %.4f status eq %.4f max iters %d" % (dd.max(), np.percentile(dd, 99.9), np.median(dd),
              (out["iters"].ravel() == ref["iters"].ravel()).mean(), (out["status"].ravel() == ref["status"].ravel()).mean(), ref["iters"].max()))
