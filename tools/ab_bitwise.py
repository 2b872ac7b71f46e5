#!/usr/bin/env python3
"""do two library builds return the same bits?  (4096 x h10 cold, 2048 x h10 fused, warm second tick)  usage: ab_bitwise.py libA.so libB.so"""
import os, subprocess, sys, hashlib
import numpy as np
if "--child" not in sys.argv:
    outs = []
    for lib in sys.argv[1:3]:
        r = subprocess.run([sys.executable, __file__, lib, "--child"], capture_output=True, text=True, timeout=150)
        outs.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
        print(os.path.basename(lib), outs[-1])
    print("IDENTICAL" if outs[0] == outs[1] else "DIFFERENT")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); pkg.engine._lib = pkg.engine.load_library(sys.argv[1])
h = hashlib.sha256()
for n, ws in ((4096, 0), (2048, 0), (512, 1)):
    sc = pkg.scenarios.config3_random_flat(nb=n)
    with pkg.Engine(pkg.make_config(sc["params"], 10, warm_start=ws), n, 0) as eng:
        for _ in range(2):
            o = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
            h.update(o["u"].tobytes()); h.update(o["iters"].tobytes()); h.update(o["grf"].tobytes())
print(h.hexdigest())
