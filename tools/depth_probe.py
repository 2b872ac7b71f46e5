"""Pipeline depth 2 / 3 / 4 through a free-running submit loop vs a loop whose slots start behind one event (what bench.py does for its timed region), with a bit-for-bit
check of the pipelined outputs against a lone handle.  A1_LIB=path: another build of the library.  Prints one JSON line."""
import json, os, sys, time, gc
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get("A1_LIB"): pkg.engine._lib = pkg.engine.load_library(os.environ["A1_LIB"])
gc.collect(); gc.disable()
n, h, steps, NB = 4096, 10, 60, 4
dev = torch.device("cuda", 0)
scs = [pkg.scenarios.config3_random_flat(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(NB)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
ds = [{k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s in scs]
with pkg.Engine(cfg, n, 0) as eng:
    ref = [eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"]) for s in scs]
res = {}
for variant in ("free_running", "one_event_start"):
    for E in (2, 3, 4):
        pipe = pkg.Pipeline(cfg, n, 0, depth=E)
        stream = torch.cuda.Stream(device=dev)
        outs = [(torch.zeros((n, 12), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
        def sub(k, after=None):
            d = ds[k % NB]; o = outs[k % NB]
            pipe.submit_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, o[1], o[2], fresh=True, after_stream=after)
        for k in range(12): sub(k)
        pipe.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if variant == "one_event_start":
            e0 = torch.cuda.Event(enable_timing=True); e0.record(stream)
            for k in range(steps): sub(k, after=stream.cuda_stream if k < E else None)
            pipe.join(stream.cuda_stream); torch.cuda.synchronize()
        else:
            for k in range(steps): sub(k)
            pipe.wait(); torch.cuda.synchronize()
        res[f"{variant}_depth{E}_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        same = all(np.array_equal(outs[b][0].cpu().numpy(), ref[b]["grf"]) and np.array_equal(outs[b][1].cpu().numpy(), ref[b]["iters"]) and np.array_equal(outs[b][2].cpu().numpy(), ref[b]["status"]) for b in range(NB))
        res[f"{variant}_depth{E}_bit_identical"] = bool(same)
        pipe.close()
print(json.dumps(res))
