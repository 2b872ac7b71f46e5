"""TEST INFRASTRUCTURE (uses oracle/): warm_start = 2 (the reference's per-tick OSQP update path) on the GPU against the oracle's restatement of OSQP's update
functions over long warm-started sequences: `robots` independent robots (each with its own carried workspace), `ticks` ticks of a trot with slowly drifting
states, periodic contact switches and occasional jumps, on the latency kernel (1 robot), the fused kernel and the split pipeline.
Usage (GPU box): python tests/tools/soak_update_path.py [ticks [horizon]] -> gpurun_out/soak_update_path[_h<horizon>].txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

pkg = g.load_package(); oracle = g.load_oracle()
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H = int(sys.argv[2]) if len(sys.argv) > 2 else 10   # (16 / 20: the quads of rows of the latency, fused and persistent kernels)
lines = []
for robots in (1, 512, 3000):
    T = ticks * (4 if robots == 1 else 1) if robots < 3000 else max(8, ticks // 6)
    rng = np.random.default_rng(robots)
    sc = pkg.scenarios.config3_random_flat(nb=robots, seed=4000 + robots, horizon=H)
    p = sc["params"]
    pr = oracle.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    carries = [oracle.update_carry(H) for _ in range(robots)]
    worst = 0.0; it_mis = 0; st_mis = 0; solves = 0; iters_sum = 0
    t0 = time.time()
    with pkg.Engine(pkg.make_config(p, H, warm_start=2), robots, 0) as eng:
        for t in range(T):
            if t > 0:
                sc["x0"][:, :12] += rng.normal(0, 1.5e-3, (robots, 12)); sc["foot"] += rng.normal(0, 5e-4, (robots, 12))
            if t % 15 == 7:      # gait switch: the stance pattern flips
                flip = rng.random(robots) < 0.5
                c = sc["contact"].copy(); c[flip] = 1 - c[flip]; c[c.sum(1) == 0] = [1, 0, 0, 1]; sc["contact"] = c
            if t % 23 == 11:     # a jump of the state (a push): the carried iterates are far from the new solution
                j = rng.random(robots) < 0.1
                sc["x0"][j, 6:12] += rng.normal(0, 0.3, (int(j.sum()), 6))
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
            for b in range(robots):
                o = oracle.mpc_solve_update(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], carries[b])
                worst = max(worst, float(np.abs(out["grf"][b] - o["grf"]).max()))
                it_mis += int(out["iters"][b] != o["info"].iters); st_mis += int(out["status"][b] != o["info"].status)
                iters_sum += o["info"].iters
            solves += robots
    lines.append(f"{robots} robot(s) x {T} ticks = {solves} solves on the update path at h = {H}: worst |dGRF| {worst:.2e} N, iteration mismatches {it_mis}, status mismatches {st_mis}, "
                 f"mean iterations {iters_sum / solves:.1f}  ({time.time() - t0:.0f} s)")
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "soak_update_path.txt" if H == 10 else "soak_update_path_h%d.txt" % H), "w").write("\n".join(lines) + "\n")
