#!/usr/bin/env python3
"""GPU probe: kernel time vs iteration count (slope = cost per ADMM iteration, intercept = set-up cost)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch

pkg = g.load_package()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
gen = {10: pkg.scenarios.config3_random_flat, 16: lambda nb: pkg.scenarios.config3_random_flat(nb=nb, horizon=16),
       20: lambda nb: pkg.scenarios.config5_divergent(nb=nb)}[H]
sc = gen(nb=n)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(sc[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")}
grf = torch.zeros((n, 12), dtype=torch.float64, device=dev)
iters = torch.zeros(n, dtype=torch.int32, device=dev); status = torch.zeros(n, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)

def run(label, reps=5, **osqp):
    cfg = pkg.make_config(sc["params"], H, warm_start=0, **osqp)
    eng = pkg.Engine(cfg, n, 0)
    ms = []
    for r in range(reps + 1):
        eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, None, iters, status, stream=st.cuda_stream)
        ms.append(eng.last_kernel_ms())
    it = iters.cpu().numpy()
    print(f"{label:28s} kernel {np.median(ms[1:]):8.3f} ms  mean_iters {it.mean():7.1f} max {it.max():5d}  nfact_mean {eng.last_nfact(n).mean():.2f}", flush=True)
    eng.close()

run("default")
run("stop@25 (eps=1e9)", eps_abs=1e9, eps_rel=1e9)
run("fixed 50 (no adapt)", eps_abs=1e-300, eps_rel=1e-300, max_iter=50, adaptive_rho=0)
run("fixed 100 (no adapt)", eps_abs=1e-300, eps_rel=1e-300, max_iter=100, adaptive_rho=0)
run("fixed 200 (no adapt)", eps_abs=1e-300, eps_rel=1e-300, max_iter=200, adaptive_rho=0)
run("fixed 100, no scaling", eps_abs=1e-300, eps_rel=1e-300, max_iter=100, adaptive_rho=0, scaling=0)
run("fixed 25, no scaling", eps_abs=1e-300, eps_rel=1e-300, max_iter=25, adaptive_rho=0, scaling=0)
