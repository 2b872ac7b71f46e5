#!/usr/bin/env python3
"""A/B of library builds at the other horizons: first-solve kernel ms at 8192 x h16 and 16384 x h20 (configs[3] / [4] shapes).  usage: ab_horizons.py libA.so libB.so ..."""
import json, os, subprocess, sys
import numpy as np
if "--child" not in sys.argv:
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, __file__, lib, "--child"], capture_output=True, text=True, timeout=200)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); pkg.engine._lib = pkg.engine.load_library(sys.argv[1])  # (load_library only caches the in-tree path)
res = {}
for gen, n, h in [c for c in (("config4_random_h16", 8192, 16), ("config5_divergent", 16384, 20)) if os.environ.get("A1_AB_H") in (None, str(c[2]))]:  # A1_AB_H=16: a slim build of one horizon
    sc = getattr(pkg.scenarios, gen)(nb=n)
    with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=0), n, 0) as eng:
        ms = []
        for _ in range(5):
            eng.set_schedule(True); eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); ms.append(eng.last_kernel_ms())
        hist = []
        for _ in range(4):
            eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); hist.append(eng.last_kernel_ms())
    res[f"h{h}"] = dict(first_ms=round(float(np.median(ms[1:])), 3), history_ms=round(float(np.median(hist[1:])), 3))
print(json.dumps(res))
