#!/usr/bin/env python3
"""How well can a first solve's queue order be guessed from the inputs alone?  (scheduling only, no result depends on it.)
TEST / ANALYSIS INFRASTRUCTURE: uses the oracle (same iteration counts and factor passes as the engine on every QP) for the true cost of each QP,
iterations + 10 x factor passes, on config3 (h10) / config4 (h16) / config5 (h20) batches of 4096 with held-out seeds, and compares
  * the shipped guess (RowSolver::predict_cost: 30 e_vz + 27 |e_vxy| + 10 [four stance legs]),
  * a gradient-boosted regressor and a tail classifier (cost >= 95th percentile) over everything the set-up kernel holds for free
    (the 12 state errors, contact bits, stance count, required vertical / horizontal acceleration, their ratio, load per leg, roll, pitch, height, feet),
by rank correlation, by recall of the long QPs among the first `rows` started, and by the makespan of a list-scheduling model of the persistent rows
(rows pull the next QP of the order when they finish one; 2048 / 1280 / 1024 rows at h = 10 / 16 / 20).  Runs on the CPU (~1 min).  Prints a table."""
import heapq, os, sys
import numpy as np
from scipy.stats import spearmanr
from sklearn.ensemble import GradientBoostingClassifier, GradientBoostingRegressor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()


def batch(gen, seed, nb=4096):
    sc = gen(nb=nb, seed=seed); p = sc["params"]; h = sc["horizon"]
    pr = O.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = O.mpc_solve_batch(pr, O.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    x0 = sc["x0"]; xref = sc["xref"].reshape(nb, h, 13); ct = sc["contact"].astype(float)
    err0 = xref[:, 0, :12] - x0[:, :12]
    ns = ct.sum(1); T = 0.0025 * h
    az = 9.8 + err0[:, 11] / T * 0.5; axy = np.hypot(err0[:, 9], err0[:, 10]) / T * 0.5
    X = np.column_stack([err0, ct, ns, az, axy, axy / np.maximum(az, 1e-3), 12 * az / ns / 180.0, x0[:, 0], x0[:, 1], x0[:, 5], sc["foot"]])
    shipped = 30 * err0[:, 11] + 27 * np.hypot(err0[:, 9], err0[:, 10]) + 10 * (ns == 4)
    return X, (ref["iters"] + 10 * ref["nfact"]).astype(float), shipped


def makespan(cost, order, rows):
    hp = [0.0] * rows; heapq.heapify(hp)
    for i in order:
        heapq.heappush(hp, heapq.heappop(hp) + cost[i])
    return max(hp)


print("family  rows | rank corr shipped / GBM | long-QP recall in the first `rows` shipped / classifier | makespan index / shipped / GBM / classifier / true order | sum/rows  longest")
for name, gen, rows, seeds in (("h10 config3", pkg.scenarios.config3_random_flat, 2048, (2001, 2002, 2003, 0xA1 + 3)),
                               ("h16 config4", pkg.scenarios.config4_random_h16, 1280, (2101, 2102, 0xA1 + 4)),
                               ("h20 config5", pkg.scenarios.config5_divergent, 1024, (2201, 2202, 0xA1 + 5))):
    tr = [batch(gen, s) for s in seeds[:-1]]; Xte, yte, shipped = batch(gen, seeds[-1])
    Xtr = np.vstack([t[0] for t in tr]); ytr = np.concatenate([t[1] for t in tr])
    reg = GradientBoostingRegressor(n_estimators=300, max_depth=4, learning_rate=0.05, subsample=0.8, random_state=0).fit(Xtr, ytr).predict(Xte)
    thr = np.quantile(ytr, 0.95)
    cls = GradientBoostingClassifier(n_estimators=200, max_depth=3, learning_rate=0.05, subsample=0.8, random_state=0).fit(Xtr, ytr >= thr).predict_proba(Xte)[:, 1]
    tail = yte >= thr
    rec = lambda p: tail[np.argsort(-p)[:rows]].sum() / tail.sum()
    n = len(yte)
    print(f"{name}  {rows} | {spearmanr(shipped, yte)[0]:.2f} / {spearmanr(reg, yte)[0]:.2f} | {rec(shipped):.2f} / {rec(cls):.2f} | "
          f"{makespan(yte, range(n), rows):.0f} / {makespan(yte, np.argsort(-shipped), rows):.0f} / {makespan(yte, np.argsort(-reg), rows):.0f} / {makespan(yte, np.argsort(-cls), rows):.0f} / "
          f"{makespan(yte, np.argsort(-yte), rows):.0f} | {yte.sum() / rows:.0f}  {yte.max():.0f}")
