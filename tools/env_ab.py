"""GPU: the same solves under two settings of one environment switch of the library (children of this script), digests of every output compared -- how the GPU suite
checks that an opt-in / A-B switch changes scheduling and nothing else.   python tools/env_ab.py VAR [n [ticks [warm_start [value_a,value_b]]]]   -> one JSON line per child"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as g
    pkg = g.load_package()
    n, ticks, warm = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sc = pkg.scenarios.config3_random_flat(nb=n)
    rng = np.random.default_rng(17)
    cfg = pkg.make_config(sc["params"], 10, warm_start=warm)
    hsh = hashlib.sha256(); ms = []
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            x0 = sc["x0"].copy(); x0[:, :12] += rng.normal(0, 0.002 * t, (n, 12))
            o = eng.solve(x0, sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
            for k in ("grf", "u", "iters", "status"):
                hsh.update(np.ascontiguousarray(o[k]).tobytes())
            ms.append(eng.last_kernel_ms())
    print("RESULT " + json.dumps({"digest": hsh.hexdigest()[:24], "kernel_ms": ms, "mean_iters": float(o["iters"].mean()), "solved": float((o["status"] == 1).mean())}))
    sys.exit(0)
var = sys.argv[1]; n = sys.argv[2] if len(sys.argv) > 2 else "4096"; ticks = sys.argv[3] if len(sys.argv) > 3 else "3"; warm = sys.argv[4] if len(sys.argv) > 4 else "0"
values = sys.argv[5].split(",") if len(sys.argv) > 5 else ["0", "1"]   # the two settings of the switch (default: off / on)
for val in values:
    r = subprocess.run([sys.executable, __file__, "child", n, ticks, warm], capture_output=True, text=True, timeout=300, env=dict(os.environ, **{var: val}))
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(json.dumps({"var": var, "value": val, **(json.loads(res[0][7:]) if res else {"error": r.stderr[-300:]})}), flush=True)
