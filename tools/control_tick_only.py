#!/usr/bin/env python3
"""Runs ON THE GPU BOX: bench.py's full_control_tick block alone (4096 robots, one C call per tick, runs of 100 ticks with the timing events on / off / off / on), three times.
usage: python tools/control_tick_only.py"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module("a1-qp-mpc-controller_amd")
for _ in range(3):
    r = bench.full_tick_probe(pkg, 0)
    print(json.dumps({k: r[k] for k in ("ms_per_tick", "ms_per_tick_with_a1mpc_set_timing_off", "last_tick_ms_by_its_own_events", "mpc_launch_ms_of_the_last_tick")}))
