#!/usr/bin/env python3
"""Throughput of the other BASELINE horizons on one GPU (configs[3] shape: h = 16, configs[4] shape: h = 20), device-resident inputs,
cold start, default OSQP settings; steady state (queue order from the previous solve) and first solve (index order).  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev)
out = {}
for name, gen, n, h in (("configs[3] shape: random flat, h=16", pkg.scenarios.config4_random_h16, 8192, 16),
                        ("configs[4] shape: mixed contacts + 30 deg pitch, h=20", pkg.scenarios.config5_divergent, 8192, 20),
                        ("configs[4] full size, h=20", pkg.scenarios.config5_divergent, 32768, 20)):
    sc = gen(nb=n)
    cfg = pkg.make_config(sc["params"], h, warm_start=0)
    d = {k: torch.from_numpy(sc[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")}
    grf = torch.zeros((n, 12), dtype=torch.float64, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev); stt = torch.zeros(n, dtype=torch.int32, device=dev)
    with pkg.Engine(cfg, n, 0) as eng:
        ms = []
        for _ in range(4):
            eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, None, it, stt, stream=st.cuda_stream)
            ms.append(eng.last_kernel_ms())
        info = eng.kernel_info() if hasattr(eng, "kernel_info") else None
    out[name] = {"batch": n, "horizon": h, "first_solve_ms": ms[0], "steady_ms": float(np.median(ms[1:])), "solves_per_s": n / (float(np.median(ms[1:])) * 1e-3),
                 "mean_iters": float(it.float().mean().item()), "max_iters": int(it.max().item()), "solved_frac": float((stt == 1).float().mean().item())}
print(json.dumps(out))
