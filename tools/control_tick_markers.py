#!/usr/bin/env python3
"""Round 6: the chained control tick (a1mpc_control_tick_device, 4096 robots, ten ticks back to back) with the handle's timing events on, off, and off with a
no-timing marker event where each timing event would have been (A1MPC_TICK_MARKERS bit mask: children of this script, one process per setting, three alternating rounds).
   python tools/control_tick_markers.py  -> one JSON line per setting"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    if sys.argv[2] == "seq":   # an explicit sequence of settings in ONE process: "seq 010101 [ticks]"
        r = bench.full_tick_probe(pkg, 0, ticks=int(sys.argv[4]) if len(sys.argv) > 4 else 10, only=[int(c) for c in sys.argv[3]])
    else:
        r = bench.full_tick_probe(pkg, 0, only=(sys.argv[2] == "on"))
    print("RESULT " + json.dumps(r)); sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "order":
    # does the POSITION of a run in the probe's alternating sequence decide its figure?  bench.py's full_tick_probe ran (off, on) x 3: off always first of a pair
    for seq, ticks in (("010101", 10), ("101010", 10), ("000000", 10), ("111111", 10), ("010101", 200), ("101010", 200)):
        r = subprocess.run([sys.executable, __file__, "child", "seq", seq, str(ticks)], capture_output=True, text=True, timeout=600)
        l = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        print(json.dumps({"sequence (1 = timing events on)": seq, "ticks_per_run": ticks, "ms_per_tick": [round(v, 4) for v in json.loads(l[0][7:])["ms_per_tick"]] if l else r.stderr[-300:]}), flush=True)
    sys.exit(0)
settings = [("on", 0), ("off", 0), ("off", 1), ("off", 2), ("off", 4), ("off", 8), ("off", 3), ("off", 15)]
res = {k: [] for k in settings}
for rnd in range(3):
    for (t, m) in settings:
        r = subprocess.run([sys.executable, __file__, "child", t], capture_output=True, text=True, timeout=300, env=dict(os.environ, A1MPC_TICK_MARKERS=str(m)))
        l = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        if l:
            res[(t, m)] += json.loads(l[0][7:])["ms_per_tick"]
for (t, m), v in res.items():
    v = sorted(v)
    print(json.dumps({"timing_events": t, "markers": m, "ms_per_tick_median": v[len(v) // 2] if v else None, "min": v[0] if v else None, "max": v[-1] if v else None, "runs": len(v)}), flush=True)
