"""GPU: stage counters of the warm-started tick (bench.warm_tick_stage_counters): 4096 robots and one robot, both warm-start semantics.
    python tools/tick_stage_probe.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = bench.graft.load_package(); pkg.load_library()
res = {}
for n in (4096, 1):
    for mode in (1, 2):
        r = bench.warm_tick_stage_counters(pkg, 0, n=n, mode=mode)
        res[f"{n}_mode{mode}"] = r
        print(n, mode, json.dumps(r))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
