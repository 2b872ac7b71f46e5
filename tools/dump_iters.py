#!/usr/bin/env python3
"""Per-QP iteration and factor-pass counts of real solves -> gpurun_out/iters_dump.npz (input of tools/wave_sim.py and of the queue-order
studies).  usage: dump_iters.py [generator horizon n ...]   default: config3_random_flat 10 4096  config3_random_flat 10 16384"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
jobs = sys.argv[1:] or ["config3_random_flat", "10", "4096", "config3_random_flat", "10", "16384"]
out = {}
for gen, h, n in zip(jobs[0::3], jobs[1::3], jobs[2::3]):
    h, n = int(h), int(n)
    sc = getattr(pkg.scenarios, gen)(nb=n)
    cfg = pkg.make_config(sc["params"], h, warm_start=0)
    key = f"{n}" if gen == "config3_random_flat" else f"{gen}_{n}"
    with pkg.Engine(cfg, n, 0) as eng:
        for rep in range(3):
            r = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
            ms = eng.last_kernel_ms()
            print(gen, n, rep, ms)
        out[f"iters_{key}"] = r["iters"]; out[f"nfact_{key}"] = eng.last_nfact(n); out[f"ms_{key}"] = ms
        eng.set_schedule(False)
        r = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); out[f"ms_index_{key}"] = eng.last_kernel_ms()
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/iters_dump.npz", **out)
