"""Why does bench.py see depth 3 slower than depth 2 when a bare submit loop does not?  Variants of the loop, one box.  (diagnostic; prints one JSON line)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n, h, steps, NB = 4096, 10, 60, 4
dev = torch.device("cuda", 0)
scs = [pkg.scenarios.config3_random_flat(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(NB)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
ds = [{k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s in scs]
res = {}
for variant in ("bare", "idle_engine", "events_join", "no_iters_status"):
    for E in (2, 3):
        extra = pkg.Engine(cfg, n, 0) if variant == "idle_engine" else None
        pipe = pkg.Pipeline(cfg, n, 0, depth=E)
        stream = torch.cuda.Stream(device=dev)
        outs = [(torch.zeros((n, 12), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
        def sub(k, after=None):
            d = ds[k % NB]; o = outs[k % NB]
            if variant == "no_iters_status": pipe.submit_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, None, None, fresh=True, after_stream=after)
            else: pipe.submit_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, o[1], o[2], fresh=True, after_stream=after)
        for k in range(12): sub(k)
        pipe.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if variant == "events_join":
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(stream)
            for k in range(steps): sub(k, after=stream.cuda_stream if k < E else None)
            pipe.join(stream.cuda_stream); e1.record(stream); torch.cuda.synchronize()
        else:
            for k in range(steps): sub(k)
            pipe.wait(); torch.cuda.synchronize()
        res[f"{variant}_depth{E}_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        pipe.close()
        if extra: extra.close()
print(json.dumps(res))
