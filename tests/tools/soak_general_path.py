"""Soak of the general path (per-step feet + per-step contact schedules) against the oracle's strided formation: [first_seed [count [n [random_settings]]]]
(random_settings = 1: every seed also draws a random combination of OSQP settings and friction / force limits, like soak_settings.py; seeds alternate between per-step feet + schedule,
a schedule alone (which the fast kernels take) and per-step feet alone)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 7000; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 6; n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
RS = len(sys.argv) > 4 and sys.argv[4] == "1"
worst = 0.0; bad = 0; tot = 0
HS = tuple(int(v) for v in os.environ["A1_SOAK_HORIZONS"].split(",")) if os.environ.get("A1_SOAK_HORIZONS") else (10, 10, 16, 20)   # (A1_SOAK_HORIZONS=4,6,8,12,14: the extended horizons)
for seed in range(lo, lo + cnt):
    h = HS[seed % len(HS)]
    sc = pkg.scenarios.config3_random_flat(nb=n, seed=seed, horizon=h, param_set=("gazebo", "hardware", "isaac")[seed % 3]); p = sc["params"]
    rng = np.random.default_rng(seed)
    vd = rng.uniform(-0.6, 0.6, (n, 1, 1, 3))
    foot = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * p["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12))
    sw = rng.integers(0, h + 1, (n, 4)); first = rng.integers(0, 2, (n, 4))
    contact = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :], 1 - first[:, None, :]).astype(np.uint8).reshape(n, h * 4))
    over = {}; fs, cs = 12, 4
    if RS:
        over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])), rho=float(10 ** rng.uniform(-2, 0.3)),
                    sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])), adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                    adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])), adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                    eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
        over["eps_rel"] = over["eps_abs"]
        p = dict(p, mu=float(rng.choice([0.3, 0.3, 0.6, 0.15])), fz_min=float(rng.choice([0.0, 0.0, 5.0])), fz_max=float(rng.choice([180.0, 120.0, 60.0])))
        kind = seed % 3
        if kind == 1: foot = np.ascontiguousarray(sc["foot"]); fs = 0          # a schedule alone: the fast kernels
        if kind == 2: contact = np.ascontiguousarray(sc["contact"]); cs = 0    # per-step feet alone
    with pkg.Engine(pkg.make_config(p, h, warm_start=0, **over), n, 0) as eng:
        out = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs)
    pr = orc.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"]); st = orc.default_settings(**over)
    dd = 0.0; mis = 0
    for b in range(n):
        r = orc.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=fs, contact_stride=cs)
        if r["info"].iters != out["iters"][b] or r["info"].status != out["status"][b]:
            mis += 1
        else:
            dd = max(dd, float(np.abs(r["grf"] - out["grf"][b]).max()))
    worst = max(worst, dd); bad += mis; tot += n
    print(seed, "h", h, ("feet+schedule", "schedule", "feet")[seed % 3] if RS else "", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in over.items()}, "max %.2e mismatches %d" % (dd, mis), flush=True)
print("TOTAL", tot, "QPs (general path), worst %.3e N, mismatching iteration counts / statuses: %d" % (worst, bad))
