"""GPU: warm-started ticks of 4096 robots with and without the previous tick's cost order (A1MPC_WARM_ORDER=0/1 children), both warm-start semantics,
and the per-tick iteration histogram that explains the difference.   python tools/warm_order_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    pkg = bench.graft.load_package(); pkg.load_library()
    out = {}
    for mode in (1, 2):
        r = bench.warm_tick_probe(pkg, 0, n=4096, ticks=24, mode=mode)
        out[f"mode{mode}"] = {"kernel_ms_per_tick": r["kernel_ms_per_tick"], "mean_iters_warm": r["mean_iters_warm"]}
    print("RESULT " + json.dumps(out))
    sys.exit(0)
for flag in ("0", "1", "0", "1"):
    env = dict(os.environ, A1MPC_WARM_ORDER=flag)
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, timeout=300, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print("A1MPC_WARM_ORDER=" + flag, line[0][7:] if line else r.stderr[-400:])
