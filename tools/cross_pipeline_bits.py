import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package()
for h in (2, 4, 6, 8, 10, 12, 14, 16, 20):
    sc = pkg.scenarios.config3_random_flat(nb=3000, seed=5200 + h, horizon=h)
    outs = {}
    with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=0), 3000, 0) as eng:
        for n in (1, 40, 600, 3000):
            outs[n] = eng.solve(sc["x0"][:n], sc["xref"][:n], sc["R"][:n], sc["foot"][:n], sc["contact"][:n], want_u=True)
        st = eng.last_stage_ms()
    line = [f"h {h} split={st[0] > 0}"]
    for n in (1, 40, 600):
        d = np.abs(outs[n]["u"] - outs[3000]["u"][:n]).max()
        line.append(f"n={n}: max|du| {d:.2e} iters_equal {np.array_equal(outs[n]['iters'], outs[3000]['iters'][:n])}")
    d = np.abs(outs[40]["u"] - outs[600]["u"][:40]).max(); line.append(f"40 vs 600: {d:.2e}")
    d = np.abs(outs[1]["u"] - outs[40]["u"][:1]).max(); line.append(f"1 vs 40: {d:.2e}")
    print(" | ".join(line), flush=True)
