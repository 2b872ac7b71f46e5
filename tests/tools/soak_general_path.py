"""Soak of the general path (per-step feet + per-step contact schedules) against the oracle's strided formation: [first_seed [count [n]]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 7000; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 6; n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
worst = 0.0; bad = 0; tot = 0
for seed in range(lo, lo + cnt):
    h = (10, 10, 16, 20)[seed % 4]
    sc = pkg.scenarios.config3_random_flat(nb=n, seed=seed, horizon=h, param_set=("gazebo", "hardware", "isaac")[seed % 3]); p = sc["params"]
    rng = np.random.default_rng(seed)
    vd = rng.uniform(-0.6, 0.6, (n, 1, 1, 3))
    foot = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * p["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12))
    sw = rng.integers(0, h + 1, (n, 4)); first = rng.integers(0, 2, (n, 4))
    contact = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :], 1 - first[:, None, :]).astype(np.uint8).reshape(n, h * 4))
    with pkg.Engine(pkg.make_config(p, h, warm_start=0), n, 0) as eng:
        out = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4)
    pr = orc.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"]); st = orc.default_settings()
    dd = 0.0; mis = 0
    for b in range(n):
        r = orc.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=12, contact_stride=4)
        if r["info"].iters != out["iters"][b] or r["info"].status != out["status"][b]:
            mis += 1
        else:
            dd = max(dd, float(np.abs(r["grf"] - out["grf"][b]).max()))
    worst = max(worst, dd); bad += mis; tot += n
    print(seed, "h", h, "max %.2e mismatches %d" % (dd, mis), flush=True)
print("TOTAL", tot, "QPs (general path), worst %.3e N, mismatching iteration counts / statuses: %d" % (worst, bad))
