#!/usr/bin/env python3
"""A/B of two builds of liba1mpc.so on the small-batch kernels (latency kernel: <= 256 QPs, fused kernel: up to the resident rows) at h = 16 / 20, where the
quads of rows replaced the twin pairs in round 4: kernel ms per launch (cold solves) and a hash of forces, full solutions, iteration counts, statuses.
usage: ab_small_batches.py libA.so libB.so"""
import hashlib, json, os, subprocess, sys
import numpy as np
if "--child" not in sys.argv:
    res = {}
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, __file__, lib, "--child"], capture_output=True, text=True, timeout=600)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:]
        print(os.path.basename(lib), line, flush=True)
        try: res[lib] = json.loads(line)
        except Exception: pass
    if len(res) == 2:
        a, b = res.values()
        print(json.dumps({"bit_identical": {k: a[k]["sha"] == b[k]["sha"] for k in a}, "kernel_ms": {k: [a[k]["ms"], b[k]["ms"]] for k in a}}))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); pkg.engine._lib = pkg.engine.load_library(sys.argv[1])
out_all = {}
for h, gen in ((16, pkg.scenarios.config4_random_h16), (20, lambda nb: pkg.scenarios.config5_divergent(nb=nb, horizon=20))):
    for n in (1, 64, 256, 1000):
        sc = gen(nb=n)
        hs = hashlib.sha256(); ms = []
        with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=0), n, 0) as eng:
            for _ in range(30 if n <= 256 else 8):
                o = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True); ms.append(eng.last_kernel_ms())
            for k in ("grf", "u", "iters", "status"): hs.update(np.ascontiguousarray(o[k]).tobytes())
        with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=2), n, 0) as eng:   # three ticks on the update path
            for _ in range(3):
                o = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
                for k in ("grf", "u", "iters", "status"): hs.update(np.ascontiguousarray(o[k]).tobytes())
        out_all["%dx%d" % (n, h)] = {"ms": round(float(np.median(ms[2:])), 4), "sha": hs.hexdigest()[:12]}
print(json.dumps(out_all))
