#!/usr/bin/env python3
"""batch-1 latency through the host-pointer ABI (config2 trot, warm start); env A1MPC_PIPELINE=fused|split selects the kernel path"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
import gc; gc.collect(); gc.disable()  # the 30-60 ms "tick 175" outlier was the Python garbage collector, not the library
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
sc = pkg.scenarios.config2_trot_sequence(n)
cfg = pkg.make_config(sc["params"], 10, warm_start=1)
for rep in range(2):
  lat = np.zeros(n); its = np.zeros(n)
  with pkg.Engine(cfg, 256, 0) as eng:
    for t in range(n):
        a = time.perf_counter()
        r = eng.solve(sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t])
        lat[t] = time.perf_counter() - a; its[t] = r["iters"][0]
    km = eng.last_kernel_ms()
  print('rep', rep, 'ticks over 2 ms:', [(t, round(lat[t]*1e3,2)) for t in range(n) if lat[t] > 2e-3][:10])
big = [(t, round(lat[t]*1e3,2), int(its[t])) for t in range(n) if lat[t] > 2e-3]
print('ticks over 2 ms:', big[:20])
lat = lat[50:] * 1e3
print(f"pipeline={os.environ.get('A1MPC_PIPELINE','auto')}: p50 {np.percentile(lat,50):.3f} ms p99 {np.percentile(lat,99):.3f} ms max {lat.max():.3f}  mean iters {its.mean():.1f}  last kernel {km:.3f} ms")
