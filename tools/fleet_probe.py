"""Closed-loop regime with TWO robot fleets on the two slots of a pipeline: each fleet's warm-started tick (fixed slot: its carried OSQP workspace lives there) runs
while the other fleet's tick drains.  ms per fleet-tick against one fleet alone on one handle.  Usage: python tools/fleet_probe.py [n [warm_mode]]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0); rng = np.random.default_rng(1)
T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
fleets = []
for f in range(2):
    a = pkg.scenarios.config3_random_flat(nb=n, seed=300 + f)
    b = {k: a[k].copy() for k in ("x0", "xref", "R", "foot", "contact")}
    b["x0"][:, :12] += rng.normal(0, 0.002, (n, 12)); b["foot"] += rng.normal(0, 0.001, (n, 12))
    fleets.append([[T(s["x0"]), T(s["xref"]), T(s["R"]), T(s["foot"]), T(s["contact"], torch.uint8)] for s in (a, b)])   # two nearby states alternate
cfg = pkg.make_config(pkg.scenarios.config3_random_flat(nb=2)["params"], 10, warm_start=mode)
outs = [(torch.zeros(n, 12, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(2)]
with pkg.Pipeline(cfg, n, 0, depth=2) as pipe:
    def tick(t, fl):
        for f in fl:
            pipe.submit_device(n, *fleets[f][t % 2], outs[f][0], None, outs[f][1], slot=f, fresh=False)
    res = {}
    for name, fl in (("one fleet alone", (0,)), ("two fleets, one per slot", (0, 1))):
        for t in range(10): tick(t, fl)
        pipe.wait(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(60): tick(t, fl)
        pipe.wait(); ms = (time.perf_counter() - t0) / (60 * len(fl)) * 1e3
        print(f"warm_start={mode}, {n} robots per fleet, {name}: {ms:.4f} ms per fleet-tick = {n / ms / 1e3:.2f} M robot-ticks/s, mean iterations {outs[0][1].float().mean().item():.1f}", flush=True)
