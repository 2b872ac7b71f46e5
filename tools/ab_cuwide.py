#!/usr/bin/env python3
"""A/B of the CU-wide ADMM kernel (five QPs per CU at h = 16) against the one-wave kernels inside ONE library: A1MPC_CU_WIDE=0 / 1 children on the same box.
usage: ab_cuwide.py lib.so [n]   -> first-solve / history kernel ms at n x h16 (default 8192) and a bit-for-bit comparison of the two runs' outputs."""
import hashlib, json, os, subprocess, sys
import numpy as np
if "--child" not in sys.argv:
    lib = sys.argv[1]; n = sys.argv[2] if len(sys.argv) > 2 else "8192"
    res = {}
    for rep in range(2):
        for mode in ("0", "1"):
            env = dict(os.environ, A1MPC_CU_WIDE=mode)
            out = subprocess.run([sys.executable, __file__, lib, n, "--child"], capture_output=True, text=True, timeout=300, env=env)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
            print("CU_WIDE=" + mode, line, flush=True)
            try: res.setdefault(mode, []).append(json.loads(line))
            except Exception: pass
    if "0" in res and "1" in res:
        print(json.dumps({"bit_identical": res["0"][0]["sha"] == res["1"][0]["sha"],
                          "first_ms": {m: [r["first_ms"] for r in res[m]] for m in res}, "history_ms": {m: [r["history_ms"] for r in res[m]] for m in res}}))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); pkg.engine._lib = pkg.engine.load_library(sys.argv[1])
n = int(sys.argv[2])
sc = pkg.scenarios.config4_random_h16(nb=n)
with pkg.Engine(pkg.make_config(sc["params"], 16, warm_start=0), n, 0) as eng:
    ms = []
    for _ in range(5):
        eng.set_schedule(True); out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); ms.append(eng.last_kernel_ms())
    hist = []
    for _ in range(4):
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); hist.append(eng.last_kernel_ms())
h = hashlib.sha256(); h.update(np.ascontiguousarray(out["grf"]).tobytes()); h.update(np.ascontiguousarray(out["iters"]).tobytes()); h.update(np.ascontiguousarray(out["status"]).tobytes())
print(json.dumps(dict(n=n, first_ms=round(float(np.median(ms[1:])), 3), history_ms=round(float(np.median(hist[1:])), 3), mean_iters=float(np.mean(out["iters"])), sha=h.hexdigest()[:16])))
