#!/usr/bin/env python3
"""Round 6: ten device-resident control ticks (a1mpc_control_tick_device, 4096 robots) with the handle's timing events on or off -- run under
rocprofv3 --hip-trace --kernel-trace to see where the tick's time goes in each setting.   python tools/control_tick_timeline.py on|off"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench
pkg = g.load_package()
print(json.dumps(bench.full_tick_probe(pkg, 0, only=(sys.argv[1] == "on"))))
