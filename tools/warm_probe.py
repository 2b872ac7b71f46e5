#!/usr/bin/env python3
"""Closed-loop regime probe: n robots, warm-started ticks (bench.py's warm_tick_probe alone), per-tick kernel time, iteration and
factor-pass statistics.  Run under `rocprofv3 --kernel-trace --stats` for the set-up / ADMM kernel split.  usage: warm_probe.py [n [ticks [horizon [warm_start_mode]]]]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 14
H = int(sys.argv[3]) if len(sys.argv) > 3 else 10
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev)
a = pkg.scenarios.config3_random_flat(nb=n, horizon=H)
rng = np.random.default_rng(5)
b = {k: a[k].copy() for k in ("x0", "xref", "R", "foot", "contact")}
b["x0"][:, :12] += rng.normal(0, 0.002, (n, 12)); b["foot"] += rng.normal(0, 0.001, (n, 12))
cfg = pkg.make_config(a["params"], H, warm_start=mode)
da = {k: torch.from_numpy(a[k]).to(dev) for k in b}; db = {k: torch.from_numpy(b[k]).to(dev) for k in b}
grf = torch.zeros((n, 12), dtype=torch.float64, device=dev)
it = torch.zeros(n, dtype=torch.int32, device=dev); stt = torch.zeros(n, dtype=torch.int32, device=dev)
with pkg.Engine(cfg, n, 0) as eng:
    for t in range(ticks):
        d = da if t % 2 == 0 else db
        eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, None, it, stt, stream=st.cuda_stream)
        ms = eng.last_kernel_ms(); i = it.cpu().numpy(); nf = eng.last_nfact(n)
        print(f"tick {t}: {ms:.3f} ms  iters mean {i.mean():.2f} max {i.max()}  hist {np.bincount(i // 25)[:8].tolist()}  factor passes mean {nf.mean():.2f} max {nf.max()}")
