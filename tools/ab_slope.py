#!/usr/bin/env python3
"""Per-iteration cost of library builds on ONE box, with repetitions (one run of tools/ab_probe.py scatters by 3-5 %): kernel ms of a batch at a fixed 1 and
a fixed K iterations (eps = 0, no rho adaptation), `reps` launches each, libraries interleaved round after round; slope = (ms(K) - ms(1)) / (K - 1) per launch
and per QP-iteration-round.  usage: ab_slope.py libA.so libB.so ... [--n 16384] [--h 10] [--k 101] [--reps 7] [--rounds 3]"""
import json, os, subprocess, sys
import numpy as np
def opt(name, d):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
n, h, K, reps, rounds = opt("--n", 16384), opt("--h", 10), opt("--k", 101), opt("--reps", 7), opt("--rounds", 3)
if "--child" not in sys.argv:
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    acc = {os.path.basename(l): {1: [], K: []} for l in libs}
    for r in range(rounds):
        for lib in libs:
            out = subprocess.run([sys.executable, __file__, lib, "--child", "--n", str(n), "--h", str(h), "--k", str(K), "--reps", str(reps)], capture_output=True, text=True, timeout=300)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception:
                print(lib, "failed:", out.stderr[-400:]); continue
            for k in (1, K): acc[os.path.basename(lib)][k] += d[str(k)]
    res = {}
    for lib, d in acc.items():
        if not d[1]: continue
        m1, mK = float(np.median(d[1])), float(np.median(d[K]))
        res[lib] = dict(ms_1=round(m1, 4), ms_K=round(mK, 4), min_1=round(min(d[1]), 4), min_K=round(min(d[K]), 4),
                        slope_us_per_iteration_launch=round((mK - m1) / (K - 1) * 1e3, 3), slope_from_mins=round((min(d[K]) - min(d[1])) / (K - 1) * 1e3, 3))
    print(json.dumps(dict(n=n, h=h, K=K, reps=reps * rounds, libs=res), indent=1))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
pkg.engine._lib = pkg.engine.load_library(sys.argv[1])
gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[h]
sc = gen(nb=n)
out = {}
for k in (1, K):
    cfg = pkg.make_config(sc["params"], h, warm_start=0, eps_abs=1e-300, eps_rel=1e-300, max_iter=k, adaptive_rho=0)
    with pkg.Engine(cfg, n, 0) as eng:
        eng.set_schedule(False)   # index order: at a fixed iteration count every QP costs the same
        ms = []
        for i in range(reps + 2):
            eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); ms.append(eng.last_kernel_ms())
    out[str(k)] = [float(x) for x in ms[2:]]
print(json.dumps(out))
