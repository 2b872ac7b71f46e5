#!/usr/bin/env python3
"""Runs ON THE GPU BOX: batch-1 latency of the general path's host entry (a1mpc_solve_batch_strided with per-step feet and a contact schedule: the interface of the reference's
test_mpc.cpp) through ctypes, warm-started.  A1MPC_ZERO_COPY_MAX=0 restores the staged path.  usage: python tools/strided_latency_probe.py [ticks]"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
gc.collect(); gc.disable()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = pkg.scenarios.config2_trot_sequence(n)
h = 10
cfg = pkg.make_config(sc["params"], h, warm_start=1)
foot = np.repeat(sc["foot"][:, None, :], h, axis=1).reshape(n, 12 * h) + 1e-3 * np.arange(h).repeat(12)[None, :]
contact = np.repeat(sc["contact"][:, None, :], h, axis=1).reshape(n, 4 * h).astype(np.uint8)
lat = np.zeros(n)
with pkg.Engine(cfg, 8, 0) as eng:
    for t in range(n):
        a = time.perf_counter(); eng.solve_strided(sc["x0"][t], sc["xref"][t], sc["R"][t], foot[t], 12, contact[t], 4); lat[t] = time.perf_counter() - a
l = lat[100:] * 1e3
print(f"general path, batch 1, ctypes, zero_copy_max={os.environ.get('A1MPC_ZERO_COPY_MAX', '8')}: p50 {np.percentile(l, 50):.4f} ms  p99 {np.percentile(l, 99):.4f} ms")
