#!/usr/bin/env python3
"""Queue order of a batch without history: the set-up kernel's cost guess vs plain index order vs history (median kernel ms of 7 solves each).
usage: first_solve_probe.py [generator horizon n ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
jobs = sys.argv[1:] or ["config3_random_flat", "10", "4096", "config3_random_flat", "10", "16384", "config4_random_h16", "16", "8192", "config5_divergent", "20", "8192"]
for gen, h, n in zip(jobs[0::3], jobs[1::3], jobs[2::3]):
    h, n = int(h), int(n)
    sc = getattr(pkg.scenarios, gen)(nb=n)
    cfg = pkg.make_config(sc["params"], h, warm_start=0)
    a = (sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    with pkg.Engine(cfg, n, 0) as eng:
        eng.solve(*a)
        res = {}
        for mode in ("guess", "index", "history"):
            ms = []
            for rep in range(7):
                if mode == "guess": eng.set_schedule(True)       # drops the history: the next solve is a "first" one
                if mode == "index": eng.set_schedule(False)
                eng.solve(*a); ms.append(eng.last_kernel_ms())
            if mode == "index": eng.set_schedule(True); eng.solve(*a)
            res[mode] = float(np.median(ms))
    print(f"{gen} h={h} n={n}: first solve ordered by the cost guess {res['guess']:.3f} ms | index order {res['index']:.3f} ms | history {res['history']:.3f} ms")
