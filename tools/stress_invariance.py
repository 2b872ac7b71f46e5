#!/usr/bin/env python3
"""Stress version of tests/test_gpu_parity.py::test_result_does_not_depend_on_position_or_history: random sub-batches (1 ... pool size,
every kernel path, both queue orders) of a pool of QPs for `seconds`; every result must be bit-identical to the QP's result in the
first, canonical solve.  usage: stress_invariance.py [seconds [pool [horizon]]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
pool = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
h = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gen = {10: "config3_random_flat", 16: "config4_random_h16", 20: "config5_divergent"}[h]
sc = getattr(pkg.scenarios, gen)(nb=pool)
cfg = pkg.make_config(sc["params"], h, warm_start=0)
rng = np.random.default_rng(2024)
args = lambda idx: (sc["x0"][idx], sc["xref"][idx], sc["R"][idx], sc["foot"][idx], sc["contact"][idx])
with pkg.Engine(cfg, pool, 0) as eng:
    ref = eng.solve(*args(np.arange(pool)), want_u=True)
    t0 = time.time(); runs = qps = 0
    while time.time() - t0 < secs:
        kind = rng.integers(0, 4)
        n = int((1, 256, 2048, pool)[kind] * rng.uniform(0.02, 1.0)) + 1
        idx = rng.choice(pool, min(n, pool), replace=False)
        eng.set_schedule(bool(rng.integers(0, 2)))
        out = eng.solve(*args(idx), want_u=True)
        ok = (out["u"] == ref["u"][idx]).all() and (out["iters"] == ref["iters"][idx]).all() and (out["status"] == ref["status"][idx]).all()
        if not ok:
            bad = np.where((out["u"] != ref["u"][idx]).any(1))[0]
            print(f"MISMATCH: batch of {len(idx)}, {len(bad)} QPs differ, first at position {bad[:5]}, max |du| {np.abs(out['u'] - ref['u'][idx]).max():.3e}")
            sys.exit(1)
        runs += 1; qps += len(idx)
print(f"h={h}: {runs} random batches, {qps} QP solves in {secs:.0f} s: all bit-identical to the canonical solve")
