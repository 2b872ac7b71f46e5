#!/usr/bin/env python3
"""Round 6: the general path with two batches in flight (a1mpc_pipeline_submit_strided_device) -- run under rocprofv3 --kernel-trace to see whether the next batch's set-up kernel
really runs in the tail of the persistent ADMM kernel.   python tools/general_pipeline_trace.py H N [depth]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
h, n = int(sys.argv[1]), int(sys.argv[2]); depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
tt = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
ins = []
for k in range(3):
    sc = pkg.scenarios.config3_random_flat(nb=n, horizon=h, seed=4242 + 31 * k)
    rk = np.random.default_rng(h + 100 * k)
    vd = rk.uniform(-0.6, 0.6, (n, 1, 1, 3))
    f = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * sc["params"]["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12))
    sw = rk.integers(0, h + 1, (n, 4)); fi = rk.integers(0, 2, (n, 4))
    c = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], fi[:, None, :], 1 - fi[:, None, :]).astype(np.uint8).reshape(n, h * 4))
    ins.append([tt(sc["x0"]), tt(sc["xref"]), tt(sc["R"]), tt(f), tt(c, torch.uint8)])
outs = [(torch.zeros(n, 12, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(3)]
with pkg.Pipeline(pkg.make_config(sc["params"], h, warm_start=0), n, 0, depth=depth) as pipe:
    def run(steps):
        for k in range(steps):
            x0_, xr_, R_, f_, c_ = ins[k % 3]; o_ = outs[k % 3]
            pipe.submit_strided_device(n, x0_, xr_, R_, f_, 12, c_, 4, o_[0], None, o_[1], o_[2], fresh=True)
        pipe.wait()
    run(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(12); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 12 * 1e3
print(json.dumps({"horizon": h, "batch": n, "depth": depth, "ms_per_batch": ms, "solves_per_s": n / ms * 1e3}))
