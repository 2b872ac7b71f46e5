#!/usr/bin/env python3
"""CPU only.  The rounding-noise band of ONE QP: how far the double-precision oracle's answer moves when one input word moves by one ulp, beside the same
experiment in x87 extended precision (tests/x87.py) and the oracle's own default-vs-exact slack.  For the tail of a parity soak: a QP where engine and oracle
part by more than the 1e-5 N bar is judged against what double precision itself leaves open on it.
usage: outlier_noise_band.py seed qp [param_set] [horizon]   (the soak's generators: scenarios.config3_random_flat(nb=4096, seed, param_set) at h = 10, config4_random_h16 / config5_divergent at 16 / 20)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
import x87


from helpers import noise_band


if __name__ == "__main__":
    seed, qp = int(sys.argv[1]), int(sys.argv[2]); ps = sys.argv[3] if len(sys.argv) > 3 else ("gazebo", "hardware", "isaac")[seed % 3]; H = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    pkg = g.load_package(); orc = g.load_oracle()
    gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[H]   # (the soak's generators)
    sc = gen(nb=4096, seed=seed, param_set=ps) if H == 10 else gen(nb=4096, seed=seed); p = sc["params"]
    pr = orc.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    base, med, mx = noise_band(orc, pr, sc, qp)
    px = x87.params(p, H)
    rx = x87.mpc_solve(px, x87.settings(), sc["x0"][qp], sc["xref"][qp], sc["R"][qp], sc["foot"][qp], sc["contact"][qp])
    rng = np.random.default_rng(1); dx = []
    for _ in range(10):
        x0 = sc["x0"][qp].copy(); j = rng.integers(0, 12); x0[j] = np.nextafter(x0[j], x0[j] + 1.0)
        r = x87.mpc_solve(px, x87.settings(), x0, sc["xref"][qp], sc["R"][qp], sc["foot"][qp], sc["contact"][qp]); dx.append(np.abs(r["grf"] - rx["grf"]).max())
    ex = orc.mpc_solve(pr, orc.exact_settings(), sc["x0"][qp], sc["xref"][qp], sc["R"][qp], sc["foot"][qp], sc["contact"][qp])
    print("seed %d QP %d (%s, h = %d): %d iterations, status %d, %d rho updates, final rho %.3e" % (seed, qp, ps, H, base["info"].iters, base["info"].status, base["info"].rho_updates, base["info"].rho_final))
    print("  double-precision oracle vs its x87 build (same iterate sequence, 2048x less rounding): %.3e N" % np.abs(base["grf"].ravel() - rx["grf"]).max())
    print("  oracle's answer under a one-ulp change of one word of x0 (40 trials): median %.2e N, max %.2e N" % (med, mx))
    print("  the x87 build under the same perturbations (10 trials): median %.2e N, max %.2e N  -> the QP is not ill-posed, double precision is noisy on it" % (np.median(dx), max(dx)))
    print("  oracle at default tolerances vs the exact optimum (eps 1e-10, %d iterations): %.3e N" % (ex["info"].iters, np.abs(ex["grf"].ravel() - base["grf"].ravel()).max()))
