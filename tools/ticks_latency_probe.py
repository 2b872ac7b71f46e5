#!/usr/bin/env python3
"""Runs ON THE GPU BOX: batch-1 latency of the tick-record host entry (a1mpc_solve_batch_ticks: what the drop-in's compute_grf calls), warm-started, through ctypes; the same ticks
through a1mpc_solve_batch for comparison.  A1MPC_ZERO_COPY_MAX=0 restores the staged path (four pageable copies in, up to four out).  usage: python tools/ticks_latency_probe.py [ticks]"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
gc.collect(); gc.disable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sc = pkg.scenarios.config2_trot_sequence(n)
cfg = pkg.make_config(sc["params"], 10, warm_start=1)
rng = np.random.default_rng(3)
tick = np.zeros((n, 22)); tick[:, 0:3] = sc["x0"][:, 0:3]; tick[:, 3:6] = sc["x0"][:, 3:6]; tick[:, 6:9] = sc["x0"][:, 6:9]; tick[:, 9:12] = sc["x0"][:, 9:12]
tick[:, 12:15] = sc["x0"][:, 0:3]; tick[:, 15] = 0.3; tick[:, 21] = 0.3
for name in ("solve_ticks", "solve"):
    lat = np.zeros(n)
    with pkg.Engine(cfg, 8, 0) as eng:
        eng.set_timing(False) if hasattr(eng, "set_timing") else None
        for t in range(n):
            a = time.perf_counter()
            if name == "solve_ticks": r = eng.solve_ticks(tick[t], sc["R"][t], sc["foot"][t], sc["contact"][t])
            else: r = eng.solve(sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t])
            lat[t] = time.perf_counter() - a
    l = lat[100:] * 1e3
    print(f"{name:12s} zero_copy_max={os.environ.get('A1MPC_ZERO_COPY_MAX', '8')} poll={os.environ.get('A1MPC_POLL_COMPLETION', '1')}: p50 {np.percentile(l, 50):.4f} ms  p99 {np.percentile(l, 99):.4f} ms  (ctypes call included)")
