#!/usr/bin/env python3
"""statistics of the largest single-block self loop of a kernel (= the ADMM hot loop) in a -save-temps listing: instructions, scratch traffic, AGPR moves, LDS reads
usage: isa_loop.py file.s kernel_substring [...]"""
import re, sys
src = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    m = re.search(r"\n(_Z\w*%s\w*):" % re.escape(key), src); i = m.start(1); j = src.index(".end_amdhsa_kernel", i)
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", src[i:j])
    best = None
    for b in blocks:
        name = b.split(":")[0]
        if re.search(r"s_cbranch\S*\s+%s\b" % re.escape(name), b):
            n = len(re.findall(r"\n\s+(?:v_|s_|ds_|scratch_|global_|buffer_)", b))
            if best is None or n > best[0]:
                best = (n, name, b)
    n, name, b = best
    c = lambda pat: len(re.findall(pat, b))
    print("%s %s: %d instructions, scratch loads %d stores %d, accvgpr reads %d writes %d, ds_read %d, v_fmac_f64_dpp %d, s_waitcnt vmcnt %d, s_nop %d" % (
        key, name, n, c(r"scratch_load"), c(r"scratch_store"), c(r"v_accvgpr_read"), c(r"v_accvgpr_write"), c(r"\bds_read"), c(r"v_fmac_f64_dpp"), c(r"s_waitcnt vmcnt"), c(r"\bs_nop")))
