#!/usr/bin/env python3
"""Profiling target: a few launches of the hot kernel on BASELINE configs[2] (4096 QPs, h=10).  usage: prof_target.py [fixed_iters [n]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
if os.environ.get("A1_LIB"):  # profile another build of the library (the engine's loader only caches the in-tree path)
    pkg.engine._lib = pkg.engine.load_library(os.environ["A1_LIB"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = 10
if os.environ.get("A1_SHAPE"):   # "n,h": another shape of BASELINE (configs[3] share 8192,16; configs[4] 32768,20; the upper batch 65536,10), first solves
    n, H = (int(x) for x in os.environ["A1_SHAPE"].split(","))
gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[H]
sc = gen(nb=n)
osqp = dict(warm_start=0)
if os.environ.get('A1_SCALING'): osqp['scaling'] = int(os.environ['A1_SCALING'])
if len(sys.argv) > 1:
    osqp.update(eps_abs=1e-300, eps_rel=1e-300, max_iter=int(sys.argv[1]), adaptive_rho=0)
    if os.environ.get('A1_ADAPT'):  # a rho update (= one more factor pass) at every checkpoint
        osqp.update(adaptive_rho=1, adaptive_rho_tolerance=1.0 + 1e-9)
cfg = pkg.make_config(sc["params"], H, **osqp)
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(sc[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")}
grf = torch.zeros((n, 12), dtype=torch.float64, device=dev)
iters = torch.zeros(n, dtype=torch.int32, device=dev); status = torch.zeros(n, dtype=torch.int32, device=dev)
eng = pkg.Engine(cfg, n, 0)
st = torch.cuda.Stream(device=dev)
for _ in range(5 if n <= 16384 else 3):
    if os.environ.get("A1_SHAPE"): eng.set_schedule(True)   # every launch a first solve
    eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, None, iters, status, stream=st.cuda_stream)
torch.cuda.synchronize()
print("kernel ms", eng.last_kernel_ms(), "mean iters", iters.float().mean().item(), "mean factor passes", float(eng.last_nfact(n).mean()))
