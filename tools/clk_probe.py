#!/usr/bin/env python3
"""reads the A1X_CLK instrumentation of a slim debug build: shader-clock ticks per iteration spent in the backward / forward sweep (one row)
usage: clk_probe.py lib.so [iters]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
pkg.engine._lib = pkg.engine.load_library(sys.argv[1])  # (load_library only caches the in-tree path)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 101
n = 16384
sc = pkg.scenarios.config3_random_flat(nb=n)
cfg = pkg.make_config(sc["params"], 10, warm_start=0, eps_abs=1e-300, eps_rel=1e-300, max_iter=k, adaptive_rho=0) if k > 0 else pkg.make_config(sc["params"], 10, warm_start=0)  # k = 0: the default settings
with pkg.Engine(cfg, n, 0) as eng:
    for _ in range(3):
        o = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    ms = eng.last_kernel_ms()
u = o["u"]
B, F, T, U, X = u[:, 0], u[:, 12], u[:, 24], u[:, 36], u[:, 48]
if k == 0:
    it = o["iters"].astype(float)
    print(json.dumps(dict(lib=os.path.basename(sys.argv[1]), kernel_ms=ms, mean_iters=float(it.mean()), cycles_per_qp=dict(iterations=float(T.mean()), factor=float(X.mean()), checks=float(U.mean())),
                          cycles_per_iteration=float(T.sum() / it.sum()), cycles_per_check=float(U.sum() / (it.sum() / 25)))))
    sys.exit(0)
print(json.dumps(dict(lib=os.path.basename(sys.argv[1]), kernel_ms=ms, iters=k, back_per_it=float(np.median(B)) / k, fwd_per_it=float(np.median(F)) / k,
                      loop_per_it=float(np.median(T)) / k, info_per_check=float(np.median(U)) / max(1, (k + 24) // 25))))
