import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
worst = []
for seed in range(8):
    n = 4096
    sc = pkg.scenarios.config3_random_flat(nb=n, seed=1000 + seed); p = sc["params"]
    cfg = pkg.make_config(p, 10, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    pr = orc.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = orc.mpc_solve_batch(pr, orc.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    dd = np.abs(out["grf"].reshape(n, 12) - ref["grf"].reshape(n, 12)).max(1)
    i = int(dd.argmax())
    print("seed", seed, "max %.2e at QP %d (iters %d, nfact %d) | p99.9 %.2e | median %.2e | same iters %.4f" % (dd.max(), i, ref["iters"].ravel()[i], ref["nfact"].ravel()[i], np.percentile(dd, 99.9), np.median(dd), (out["iters"].ravel() == ref["iters"].ravel()).mean()))
