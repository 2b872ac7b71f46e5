"""Do back-to-back first solves of DISTINCT batches overlap when they are issued through E handles on E streams?
The tail of a 4096-QP launch (its few 150-225-iteration QPs) leaves most of the chip idle for the last 0.15-0.25 ms; the next batch's
set-up kernel and persistent rows could run there.  Prints ms per batch for E = 1, 2, 3 (same box, same batches, every step a first solve).
Usage: python tools/overlap_probe.py [n [h [steps]]]"""
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
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
NB = 4
dev = torch.device("cuda", 0)
mk = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[h]
scs = [mk(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(NB)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
ds = [{k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s in scs]
res = {"n": n, "h": h, "steps": steps}
ref_grf = None
for E in (1, 2, 3, 1, 2):
    engs = [pkg.Engine(cfg, n, 0) for _ in range(E)]
    sts = [torch.cuda.Stream(device=dev) for _ in range(E)]
    outs = [(torch.zeros((n, 12), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
            for _ in range(max(E, NB))]

    def step(k):
        e = k % E; d = ds[k % NB]; o = outs[k % len(outs)]
        engs[e].set_schedule(True)
        engs[e].solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], o[0], None, o[1], o[2], stream=sts[e].cuda_stream)

    for k in range(2 * E + 2):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # results must not depend on how many handles were in flight
    step(0); torch.cuda.synchronize()
    grf0 = outs[0][0].cpu().numpy().copy()
    if ref_grf is None:
        ref_grf = grf0
    same = bool((grf0 == ref_grf).all())
    res.setdefault(f"E{E}", []).append({"ms_per_batch": ms, "solves_per_s": n / (ms * 1e-3), "bit_identical_to_E1": same})
    print(f"E={E}: {ms:.4f} ms per batch, {n / (ms * 1e-3) / 1e6:.3f} M solves/s, identical={same}", flush=True)
    for e in engs:
        e.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/overlap_probe_{n}_h{h}.json", "w"), indent=1)
