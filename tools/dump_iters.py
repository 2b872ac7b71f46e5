import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package()
out = {}
for n in (4096, 16384):
    sc = pkg.scenarios.config3_random_flat(nb=n)
    cfg = pkg.make_config(sc["params"], 10, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        for rep in range(3):
            r = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
            ms = eng.last_kernel_ms()
            print(n, rep, ms)
        out[f"iters_{n}"] = r["iters"]; out[f"nfact_{n}"] = eng.last_nfact(n); out[f"ms_{n}"] = ms
        eng.set_schedule(False)
        r = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); out[f"ms_index_{n}"] = eng.last_kernel_ms()
np.savez("gpurun_out/iters_dump.npz", **out)
