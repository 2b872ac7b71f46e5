#!/usr/bin/env python3
"""The bench's timed region on the pipeline alone: K first solves of 4096 x h10 through a depth-2 pipeline whose slots all start behind one event (what bench.py does),
for K = 20 / 40 / 80, repeated; HIP-event and host-clock ms per batch.  Separates the steady-state rate from the fixed ramp-up / drain cost of a timed region.
usage: pipe_start_probe.py [lib.so]   (environment: whatever experiment switches the library reads, e.g. A1MPC_PIPE_PRIORITY / A1MPC_PIPE_STAGGER)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
if len(sys.argv) > 1:
    pkg.engine._lib = pkg.engine.load_library(sys.argv[1])
n, h, NB = 4096, 10, 4
dev = torch.device("cuda", 0)
scs = [pkg.scenarios.config3_random_flat(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(NB)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
ds = [{k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s in scs]
outs = [(torch.zeros((n, 12), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
stream = torch.cuda.Stream(device=dev)
depth = int(os.environ.get("A1_DEPTH", "2"))
with pkg.Pipeline(cfg, n, 0, depth=depth) as pipe:
    def submit(k, after=None):
        d = ds[k % NB]; o = outs[k % NB]
        pipe.submit_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, o[1], o[2], fresh=True, after_stream=after)
    for k in range(13):
        submit(k)
    pipe.wait(); torch.cuda.synchronize()
    for steps in (20, 20, 20, 40, 40, 80):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(stream)
        for k in range(steps):
            submit(k, after=stream.cuda_stream if k < depth else None)
        pipe.join(stream.cuda_stream); e1.record(stream); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3; ev = e0.elapsed_time(e1) / steps
        print(f"steps {steps:3d}: {ev:.4f} ms per batch by HIP events ({ev * steps:.2f} ms total), {wall:.4f} by the host clock, {n / wall / 1e3:.3f} M solves/s", flush=True)
