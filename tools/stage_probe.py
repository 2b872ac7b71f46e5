#!/usr/bin/env python3
"""stage split of a launch (a1mpc_last_stage_ms: formation + Ruiz (+ queue order) | factor + iterate) at a few batch sizes, first solve and history order
usage: stage_probe.py [lib.so]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if len(sys.argv) > 1:
    pkg.engine._lib = pkg.engine.load_library(sys.argv[1])  # (load_library only caches the in-tree path)
out = {}
for h, gen, sizes in ((10, "config3_random_flat", (4096, 16384, 65536)), (16, "config4_random_h16", (8192,)), (20, "config5_divergent", (16384,))):
    for n in sizes:
        sc = getattr(pkg.scenarios, gen)(nb=n)
        with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=0), n, 0) as eng:
            first = []
            for _ in range(4):
                eng.set_schedule(True); eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); first.append(eng.last_stage_ms())
            hist = []
            for _ in range(4):
                eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); hist.append(eng.last_stage_ms())
        f = np.median(np.array(first[1:]), axis=0); hh = np.median(np.array(hist[1:]), axis=0)
        out[f"h{h}_{n}"] = dict(first_setup_ms=round(float(f[0]), 4), first_admm_ms=round(float(f[1]), 4), history_setup_ms=round(float(hh[0]), 4), history_admm_ms=round(float(hh[1]), 4))
print(json.dumps(out, indent=1))
