#!/usr/bin/env python3
"""usage: gpu_probe2.py H batch [key=value ...]  -> kernel ms for default settings + overrides"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
H = int(sys.argv[1]); n = int(sys.argv[2])
over = {}
for kv in sys.argv[3:]:
    k, v = kv.split("="); over[k] = float(v) if "." in v or "e" in v else int(v)
gen = {1: None, 10: pkg.scenarios.config3_random_flat, 16: lambda nb: pkg.scenarios.config3_random_flat(nb=nb, horizon=16, seed=0xA1 + 4),
       20: lambda nb: pkg.scenarios.config5_divergent(nb=nb)}[H]
sc = gen(nb=n)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(sc[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")}
grf = torch.zeros((n, 12), dtype=torch.float64, device=dev)
iters = torch.zeros(n, dtype=torch.int32, device=dev); status = torch.zeros(n, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
cfg = pkg.make_config(sc["params"], H, warm_start=0, **over)
eng = pkg.Engine(cfg, n, 0)
ms = []
for r in range(6):
    eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, None, iters, status, stream=st.cuda_stream)
    ms.append(eng.last_kernel_ms())
it = iters.cpu().numpy()
m = float(np.median(ms[1:]))
print(f"H={H} n={n} rows/wg={eng.kernel_info()['qps_per_workgroup']} {over}: kernel {m:.3f} ms  {n / m / 1e3:.3f} M solves/s  mean_iters {it.mean():.1f} max {it.max()} solved {(status.cpu().numpy() == 1).mean():.3f}", flush=True)
