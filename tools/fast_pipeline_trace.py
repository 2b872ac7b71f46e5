#!/usr/bin/env python3
"""Round 6: the fast path's pipeline at a given depth in a free-running submit loop -- run under rocprofv3 --kernel-trace to see WHEN the set-up kernel of a batch runs and when its
persistent kernel starts (set-up ahead: a1mpc_pipeline with three slots).   python tools/fast_pipeline_trace.py [depth [n [h [steps]]]]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; h = int(sys.argv[3]) if len(sys.argv) > 3 else 10
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 24
NB = 4
dev = torch.device("cuda", 0)
scs = [pkg.scenarios.config3_random_flat(nb=n, seed=0xA1 + 3 + 17 * k, horizon=h) for k in range(NB)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
ds = [{k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s in scs]
outs = [(torch.zeros((n, 12), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
with pkg.Pipeline(cfg, n, 0, depth=depth) as pipe:
    def run(k_):
        for k in range(k_):
            d = ds[k % NB]; o = outs[k % NB]
            pipe.submit_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, o[1], o[2], fresh=True)
        pipe.wait(); torch.cuda.synchronize()
    run(16)
    t0 = time.perf_counter(); run(steps); ms = (time.perf_counter() - t0) / steps * 1e3
print(json.dumps({"depth": depth, "batch": n, "horizon": h, "ms_per_batch": ms, "solves_per_s": n / ms * 1e3, "gate": os.environ.get("A1MPC_PIPELINE_GATE", "1")}))
