"""Soak: the oracle's two linear-system back ends (reduced dense Cholesky vs LDL' of the full quasi-definite KKT matrix, QDLDL-style) on
the same QPs -- iteration counts / statuses must be identical, the force difference is the yardstick for what an algebraically
equivalent solver changes.  TEST INFRASTRUCTURE (uses oracle/).  python tests/tools/linsys_soak.py [n_h10 n_h16 n_h20] -> profiles/r02_linsys_soak.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import __graft_entry__ as g
from helpers import oracle_batch

O = g.load_oracle(); S = g.load_package().scenarios
counts = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [90000, 6000, 4000]
nthreads = int(os.environ.get("SOAK_THREADS", "6"))
plan = [("config3_random_flat", 10, counts[0]), ("config3_random_flat", 16, counts[1]), ("config5_divergent", 20, counts[2])]
res = dict(qps=0, iter_mismatch=0, status_mismatch=0, nfact_mismatch=0, max_du_N=0.0, per=[])
t0 = time.time()
for gen, h, total in plan:
    done = 0; seed = 1000 + h
    while done < total:
        nb = min(2048, total - done)
        for ps in ("gazebo", "hardware", "isaac"):
            sc = getattr(S, gen)(nb=nb, seed=seed, horizon=h, param_set=ps); seed += 1
            pr_kw = dict(want_u=True)
            a = O.mpc_solve_batch(__import__("helpers").oracle_params(O, sc), O.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], nthreads=nthreads, want_u=True)
            b = O.mpc_solve_batch(__import__("helpers").oracle_params(O, sc), O.default_settings(linsys=1), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], nthreads=nthreads, want_u=True)
            res["qps"] += nb; res["iter_mismatch"] += int((a["iters"] != b["iters"]).sum()); res["status_mismatch"] += int((a["status"] != b["status"]).sum())
            res["nfact_mismatch"] += int((a["nfact"] != b["nfact"]).sum())
            same = a["iters"] == b["iters"]
            du = float(np.abs(a["u"] - b["u"])[same].max()); res["max_du_N"] = max(res["max_du_N"], du)
            res["per"].append(dict(gen=gen, h=h, param_set=ps, n=nb, max_du_N=du, iter_mismatch=int((~same).sum())))
            done += nb
            if done >= total: break
        print(h, done, res["qps"], res["iter_mismatch"], res["max_du_N"], round(time.time() - t0), flush=True)
res["seconds"] = time.time() - t0
json.dump(res, open(os.path.join(ROOT, "profiles", "r02_linsys_soak.json"), "w"), indent=1)
print({k: v for k, v in res.items() if k != "per"})
