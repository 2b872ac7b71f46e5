import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
worst = 0; bad = 0; tot = 0
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 3000; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 12  # usage: soak_parity.py [first_seed [count [horizon]]]
H = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[H]
for seed in range(lo, lo + cnt):
    n = 4096
    sc = gen(nb=n, seed=seed, param_set=("gazebo", "hardware", "isaac")[seed % 3]) if H == 10 else gen(nb=n, seed=seed); p = sc["params"]
    cfg = pkg.make_config(p, H, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        out2 = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    assert np.array_equal(out["grf"], out2["grf"])
    pr = orc.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = orc.mpc_solve_batch(pr, orc.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    dd = np.abs(out["grf"].reshape(n, 12) - ref["grf"].reshape(n, 12)).max(1)
    same = (out["iters"].ravel() == ref["iters"].ravel()); steq = (out["status"].ravel() == ref["status"].ravel())
    worst = max(worst, dd.max()); bad += int((~same).sum() + (~steq).sum()); tot += n
    print(seed, "max %.2e same iters %.5f status eq %.5f" % (dd.max(), same.mean(), steq.mean()), flush=True)
print("TOTAL h =", H, ":", tot, "QPs, worst %.3e N, mismatching iteration counts / statuses: %d" % (worst, bad))
