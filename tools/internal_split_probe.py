import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
import torch
pkg = g.load_package()
h, n = int(sys.argv[1]), int(sys.argv[2]); parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2; general = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
tt = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
sc = pkg.scenarios.config3_random_flat(nb=n, horizon=h, seed=4242)
rk = np.random.default_rng(h)
if general:
    vd = rk.uniform(-0.6, 0.6, (n, 1, 1, 3))
    f = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * sc["params"]["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12)); fs = 12
    sw = rk.integers(0, h + 1, (n, 4)); fi = rk.integers(0, 2, (n, 4))
    c = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], fi[:, None, :], 1 - fi[:, None, :]).astype(np.uint8).reshape(n, h * 4)); cs = 4
else:
    f = sc["foot"]; fs = 0; c = sc["contact"]; cs = 0
X0, XR, R_, F_, C_ = tt(sc["x0"]), tt(sc["xref"]), tt(sc["R"]), tt(f), tt(c, torch.uint8)
grf = torch.zeros(n, 12, dtype=torch.float64, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
cfg = pkg.make_config(sc["params"], h, warm_start=0)
res = {}
with pkg.Pipeline(cfg, n, 0, depth=1) as lone:
    def run_lone():
        lone.submit_strided_device(n, X0, XR, R_, F_, fs, C_, cs, grf, None, it, st, fresh=True); lone.wait()
    for _ in range(3): run_lone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run_lone()
    res["lone_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    ref = (grf.cpu().numpy().copy(), it.cpu().numpy().copy())
m = n // parts
with pkg.Pipeline(cfg, m + parts, 0, depth=parts) as pipe:
    def run_split():
        for k in range(parts):
            sl = slice(k * m, (k + 1) * m if k < parts - 1 else n); cnt = sl.stop - sl.start
            pipe.submit_strided_device(cnt, X0[sl], XR[sl], R_[sl], F_[sl], fs, C_[sl], cs, grf[sl], None, it[sl], st[sl], slot=k, fresh=True)
        pipe.wait()
    for _ in range(3): run_split()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run_split()
    res["split_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    res["same_bits"] = bool(np.array_equal(grf.cpu().numpy(), ref[0]) and np.array_equal(it.cpu().numpy(), ref[1]))
res.update(h=h, n=n, parts=parts, general=general, lone_Msolves=n / res["lone_ms"] / 1e3, split_Msolves=n / res["split_ms"] / 1e3)
print(json.dumps(res))
