#!/usr/bin/env python3
"""Prints the timeline of the last launches in a rocprofv3 kernel-trace CSV: per kernel start / end relative to the first listed, stream / queue, duration.
   python tools/trace_timeline.py kernel_trace.csv [last_n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = [r for r in rows if "a1mpc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void a1mpc::", "").replace("a1mpc_", "")[:40]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:9.3f} -> {e:9.3f} ms  ({e - s:6.3f})  q{r.get('Queue_Id', '?'):>3}  {name}")
