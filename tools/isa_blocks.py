#!/usr/bin/env python3
"""Per-basic-block instruction statistics of a gfx950 .s file (hipcc -save-temps): finds spills in hot loops.
usage: isa_blocks.py file.s kernel_substring [min_instr]"""
import re
import sys

src = open(sys.argv[1]).read()
key = sys.argv[2]
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = re.search(r"\n(_Z\w*%s\w*):" % re.escape(key), src)
i = m.start(1)
j = src.index(".end_amdhsa_kernel", i)
blocks = []
cur = None
for l in src[i:j].split("\n"):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        cur = dict(name=mm.group(1), n=0, sld=0, sst=0, f64=0, dpp=0, ds=0, vmem=0, br=[])
        blocks.append(cur)
        continue
    if cur is None or not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"):
        continue
    cur["n"] += 1
    cur["sld"] += "scratch_load" in l
    cur["sst"] += "scratch_store" in l
    cur["f64"] += bool(re.search(r"_f64", l))
    cur["dpp"] += "_dpp" in l
    cur["ds"] += bool(re.search(r"\bds_", l))
    cur["vmem"] += bool(re.search(r"global_|buffer_", l))
    b = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if b:
        cur["br"].append(b.group(1))
tot = dict(n=0, sld=0, sst=0)
for b in blocks:
    for k in tot:
        tot[k] += b[k]
    if b["n"] >= mi:
        print("%-10s n=%5d sld=%4d sst=%4d f64=%4d dpp=%4d ds=%4d vmem=%3d -> %s" % (b["name"], b["n"], b["sld"], b["sst"], b["f64"], b["dpp"], b["ds"], b["vmem"], ",".join(b["br"])))
print("total", tot)
