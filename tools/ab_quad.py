#!/usr/bin/env python3
"""A/B of the quad-of-rows ADMM kernels (h = 20: the one-wave kernel; h = 16: waves 1-3 of the CU-wide kernel) against the twin-pair kernels inside ONE library:
A1MPC_QUAD=0 / 1 children on the same box.  usage: ab_quad.py lib.so horizon [n] [warm]  -> first-solve / history kernel ms and a bit-for-bit comparison of the two runs'
outputs (forces, the full solution, iteration counts, statuses; warm = 1: three warm-started ticks on top, their outputs hashed too)."""
import hashlib, json, os, subprocess, sys
import numpy as np
if "--child" not in sys.argv:
    lib = sys.argv[1]; hz = sys.argv[2]; n = sys.argv[3] if len(sys.argv) > 3 else ("32768" if hz == "20" else "8192"); warm = sys.argv[4] if len(sys.argv) > 4 else "0"
    res = {}
    for rep in range(2):
        for mode in ("0", "1"):
            env = dict(os.environ, A1MPC_QUAD=mode)
            out = subprocess.run([sys.executable, __file__, lib, hz, n, warm, "--child"], capture_output=True, text=True, timeout=600, env=env)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:]
            print("QUAD=" + mode, line, flush=True)
            try: res.setdefault(mode, []).append(json.loads(line))
            except Exception: pass
    if "0" in res and "1" in res:
        print(json.dumps({"horizon": int(hz), "n": int(n), "bit_identical": res["0"][0]["sha"] == res["1"][0]["sha"],
                          "first_ms": {m: [r["first_ms"] for r in res[m]] for m in res}, "history_ms": {m: [r["history_ms"] for r in res[m]] for m in res}}))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); pkg.engine._lib = pkg.engine.load_library(sys.argv[1])
hz = int(sys.argv[2]); n = int(sys.argv[3]); warm = sys.argv[4] == "1"
sc = pkg.scenarios.config4_random_h16(nb=n) if hz == 16 else pkg.scenarios.config5_divergent(nb=n, horizon=hz)
h = hashlib.sha256()
def add(out):
    for k in ("grf", "u", "iters", "status"):
        if k in out and out[k] is not None: h.update(np.ascontiguousarray(out[k]).tobytes())
with pkg.Engine(pkg.make_config(sc["params"], hz, warm_start=0), n, 0) as eng:
    ms = []
    for _ in range(5):
        eng.set_schedule(True); out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True); ms.append(eng.last_kernel_ms())
    add(out)
    hist = []
    for _ in range(4):
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True); hist.append(eng.last_kernel_ms())
    add(out)
if warm:
    with pkg.Engine(pkg.make_config(sc["params"], hz, warm_start=1), n, 0) as eng:
        for _ in range(3):
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True); add(out)
print(json.dumps(dict(n=n, first_ms=round(float(np.median(ms[1:])), 3), history_ms=round(float(np.median(hist[1:])), 3), mean_iters=float(np.mean(out["iters"])), sha=h.hexdigest()[:16])))
