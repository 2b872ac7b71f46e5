#!/usr/bin/env python3
"""Kernel time of the general path (per-step feet + per-step contact schedules, a1mpc_solve_batch_strided) beside the fast path on the same
states.  -> gpurun_out/r02/general_path_probe.json"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.scenarios
out = []
for h, n in ((10, 4096), (10, 16384), (16, 8192), (20, 8192)):
    sc = S.config3_random_flat(nb=n, horizon=h)
    rng = np.random.default_rng(h)
    vd = rng.uniform(-0.6, 0.6, (n, 1, 1, 3))
    foot = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * sc["params"]["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12))
    sw = rng.integers(0, h + 1, (n, 4)); first = rng.integers(0, 2, (n, 4))
    contact = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :], 1 - first[:, None, :]).astype(np.uint8).reshape(n, h * 4))
    cfg = pkg.make_config(sc["params"], h, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        ms_f, ms_g, ms_s = [], [], []
        for _ in range(3):
            eng.set_schedule(True)
            a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); ms_f.append(eng.last_kernel_ms())
            b = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4); ms_g.append(eng.last_kernel_ms())
            eng.set_schedule(True)   # contact schedule alone (feet step-invariant): the fast kernels
            c = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], sc["foot"], 0, contact, 4); ms_s.append(eng.last_kernel_ms())
    out.append(dict(horizon=h, batch=n, fast_path_kernel_ms=float(np.median(ms_f)), general_path_kernel_ms=float(np.median(ms_g)),
                    fast_solves_per_s=n / (float(np.median(ms_f)) * 1e-3), general_solves_per_s=n / (float(np.median(ms_g)) * 1e-3),
                    schedule_only_kernel_ms=float(np.median(ms_s)), schedule_only_solves_per_s=n / (float(np.median(ms_s)) * 1e-3), mean_iters_schedule_only=float(c["iters"].mean()),
                    solved_schedule_only=float((c["status"] == 1).mean()), mean_iters_fast=float(a["iters"].mean()), mean_iters_general=float(b["iters"].mean()), solved_general=float((b["status"] == 1).mean())))
    print(out[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/general_path_probe.json", "w"), indent=1)
