#!/usr/bin/env python3
"""Kernel-tuning helper: solve BASELINE configs[2] (4096 QPs, h = 10) on the GPU, compare with the oracle, print the kernel time.
usage: quick_parity.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = pkg.scenarios.config3_random_flat(nb=n); p = sc["params"]
cfg = pkg.make_config(p, 10, warm_start=0)
with pkg.Engine(cfg, n, 0) as eng:
    for _ in range(3):
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        ms = eng.last_kernel_ms()
pr = orc.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
ref = orc.mpc_solve_batch(pr, orc.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
grf = out["grf"] if isinstance(out, dict) else out[0]
it = out["iters"] if isinstance(out, dict) else out[1]
print("kernel ms %.4f | max |grf - oracle| = %.3e N | same iters %.4f | mean iters %.2f | status ok %.4f" % (
    ms, np.abs(np.asarray(grf).reshape(n, 12) - ref["grf"].reshape(n, 12)).max(), (np.asarray(it).ravel() == ref["iters"].ravel()).mean(),
    np.asarray(it).mean(), (np.asarray(out["status"]).ravel() == ref["status"].ravel()).mean()))
