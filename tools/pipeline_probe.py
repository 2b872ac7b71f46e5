"""a1mpc_pipeline: ms per batch by depth and by the number of batches in the timed region (ramp-up / drain of the pipeline), first solves of distinct batches.
Usage: python tools/pipeline_probe.py [n [h]]"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NB = 4
dev = torch.device("cuda", 0)
mk = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[h]
scs = [mk(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(NB)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
ds = [{k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s in scs]
outs = [(torch.zeros((n, 12), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
res = {"n": n, "h": h, "rows": []}
stream = torch.cuda.Stream(device=dev)
for depth in (1, 2, 3, 4):
    with pkg.Pipeline(cfg, n, 0, depth=depth) as pipe:
        def submit(k, after=None):
            d = ds[k % NB]; o = outs[k % NB]
            pipe.submit_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, o[1], o[2], fresh=True, after_stream=after)
        for steps in (10, 20, 40, 80):
            for k in range(4):
                submit(k)
            pipe.wait(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record(stream)
            for k in range(steps):
                submit(k, after=stream.cuda_stream if k < depth else None)
            pipe.join(stream.cuda_stream); e1.record(stream); torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps * 1e3; ev = e0.elapsed_time(e1) / steps
            res["rows"].append({"depth": depth, "steps": steps, "event_ms_per_batch": ev, "wall_ms_per_batch": wall})
            print(f"depth {depth} steps {steps:3d}: {ev:.4f} ms per batch by HIP events, {wall:.4f} by the host clock, {n / ev / 1e3:.3f} M solves/s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/pipeline_probe_{n}_h{h}.json", "w"), indent=1)
