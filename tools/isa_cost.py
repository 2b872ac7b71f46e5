#!/usr/bin/env python3
"""Weighted issue-cost estimate of the largest self-loop block of a kernel (the ADMM hot loop) from a -save-temps .s file.
Costs (shader cycles, one wave per SIMD) come from tools/ubench/issue_cost_ubench.hip + f64_dpp_ubench.hip on an MI355X.
usage: isa_cost.py file.s kernel_substring"""
import re, sys, collections
src = open(sys.argv[1]).read(); key = sys.argv[2]
m = re.search(r"\n(_Z\w*%s\w*):" % re.escape(key), src)
body = src[m.start(1):src.index(".end_amdhsa_kernel", m.start(1))]
blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
best = None
for b in blocks:
    name = b.split(":")[0]
    ins = [l.split() for l in b.split("\n") if re.match(r"\s*(v_|s_|ds_|global_|buffer_|scratch_|flat_)", l)]  # also the lines of multi-instruction asm blocks
    if any(re.search(r"s_cbranch\S*\s+%s\b" % re.escape(name), " ".join(i)) for i in ins):
        if best is None or len(ins) > len(best[1]): best = (name, ins)
name, ins = best
def cost(i):
    op = i[0]
    if op == "v_fmac_f64_dpp": return 6.3
    if op.startswith("ds_"): return 10.0
    if op == "s_waitcnt": return 4.0
    if op == "s_nop": return 4.0 * (int(i[1]) + 1)
    if op.startswith("v_accvgpr"): return 4.0
    if "_f64" in op: return 6.0
    if op.startswith("s_"): return 4.0
    return 7.0
cnt = collections.Counter(i[0] for i in ins); tot = sum(cost(i) for i in ins)
print("%s: %d instructions, estimated %.0f cycles per trip" % (name, len(ins), tot))
print("  " + "  ".join("%s=%d" % kv for kv in cnt.most_common(16)))
