import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package()
for n in (1, 4, 8, 16, 32, 128, 256):
    sc = pkg.scenarios.config3_random_flat(nb=n); cfg = pkg.make_config(sc["params"], 10, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        ms = []
        for _ in range(8):
            eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); ms.append(eng.last_kernel_ms())
    print(os.environ.get("A1MPC_COOP_SETUP", "1"), n, "%.4f" % np.median(ms[2:]))
