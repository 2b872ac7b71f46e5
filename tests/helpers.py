"""Shared helpers of the test-suite (oracle drivers, tolerances)."""
import numpy as np

# Parity bar (DESIGN.md "Parity"): the GPU runs the SAME algorithm as the oracle (OSQP 0.6 restated), so on
# identical inputs and settings it must terminate at the same iteration with the same status and agree in the
# returned forces to round-off.  The tolerance is absolute, in Newton, on forces of 1..180 N.
TOL_FORCE_N = 1e-5          # ||u_gpu - u_oracle||_inf, same settings, same iteration count; observed over 8 x 4096 random QPs: median 1.5e-12,
                            # 99.9 % < 5e-9, worst 1.4e-6 N (most of the worst-case gap is the double-precision oracle's own rounding: DESIGN.md 5)
TOL_FORCE_ANY_BATCH_N = 1e-5   # the same bound for arbitrary random batches (seeds no other test uses)
TOL_FORCE_BALANCE_N = 2e-4  # balance QP: P = 1e-3 I + M'QM, cond ~ 1e6 -- round-off is amplified more (observed <= 2e-5 N)
MIN_SAME_ITERS = 1.0        # EVERY problem must stop at the oracle's iteration with the oracle's status (round 6, VERDICT r5 item 7: not one mismatch has been
                            # observed in > 2 M fresh QPs at OSQP's default tolerances, so the gate is what is observed).  The only callers that pass a smaller
                            # fraction are named exceptions with their reason beside them: test_exact_mode_h10 (eps 1e-10: the termination threshold itself
                            # lies inside double-precision round-off of the residuals).  The exact-mode resolver below stays as a DIAGNOSTIC for those.


def oracle_params(O, sc):
    p = sc["params"]
    return O.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])


def oracle_batch(O, sc, n=None, settings=None, want_u=True):
    n = len(sc["x0"]) if n is None else n
    st = settings if settings is not None else O.default_settings()
    return O.mpc_solve_batch(oracle_params(O, sc), st, sc["x0"][:n], sc["xref"][:n], sc["R"][:n], sc["foot"][:n], sc["contact"][:n],
                             want_u=want_u)


def take(sc, n):
    out = dict(sc)
    for k in ("x0", "xref", "R", "foot", "contact"):
        out[k] = sc[k][:n]
    return out


def exact_resolver(O, sc, settings=None):
    """for compare(): solves the given QPs of `sc` with the oracle in exact mode (eps 1e-10) and in the default mode -> (u_exact, u_default), each (k, 12 h)"""
    def resolve(idx):
        sub = dict(sc)
        for k in ("x0", "xref", "R", "foot", "contact"):
            sub[k] = sc[k][idx]
        over = {} if settings is None else {k: getattr(settings, k) for k in ("rho", "sigma", "alpha", "scaling", "adaptive_rho", "adaptive_rho_interval", "adaptive_rho_tolerance", "check_termination")}
        ex = oracle_batch(O, sub, settings=O.exact_settings(**over))
        return ex["u"]
    return resolve


def compare(out, ref, tol=TOL_FORCE_N, min_same=MIN_SAME_ITERS, resolve=None):
    """GPU vs oracle on the same QPs.  QPs that stopped at the oracle's iteration: same status, forces within `tol`.  QPs with ANOTHER iteration count (a termination
    test within round-off of its threshold may flip; at most 1 - min_same of the batch) are not dropped: with `resolve` (exact_resolver) each is compared against the
    oracle's exact-mode optimum and must lie within 3 x the distance the default-tolerance oracle itself keeps from it on the same QP (the flip is one
    25-iteration checkpoint earlier or later) -- OSQP's own slack at its default tolerances, SURVEY 7.2(1).  Without `resolve` any such QP fails the comparison."""
    same = out["iters"] == ref["iters"]
    frac = same.mean()
    assert frac >= min_same, f"only {frac:.4f} of the problems stopped at the oracle's iteration"
    assert (out["status"][same] == ref["status"][same]).all()
    du = np.abs(out["u"] - ref["u"])[same].max() if out.get("u") is not None else 0.0
    dg = np.abs(out["grf"] - ref["grf"])[same].max()
    assert du <= tol and dg <= tol, (du, dg)
    worst_ratio = 0.0
    if not same.all():
        idx = np.flatnonzero(~same)
        assert resolve is not None, f"{len(idx)} QP(s) with another iteration count than the oracle's and no exact-mode resolver to check them against: {idx[:8]}"
        assert out.get("u") is not None and ref.get("u") is not None, "resolving iteration mismatches needs the full-horizon forces (want_u)"
        ue = resolve(idx)
        for j, i in enumerate(idx):
            slack = np.abs(ref["u"][i] - ue[j]).max()     # what the default tolerances leave open on this QP (oracle default vs oracle exact)
            d = np.abs(out["u"][i] - ue[j]).max()
            assert d <= 3.0 * slack + tol, f"QP {i}: iterations {out['iters'][i]} vs {ref['iters'][i]}; {d:.3e} N from the exact optimum, the oracle's own default-mode slack is {slack:.3e} N"
            worst_ratio = max(worst_ratio, d / max(slack, 1e-300))
    return dict(same_frac=float(frac), du=float(du), dgrf=float(dg), resolved=int((~same).sum()), worst_resolved_ratio=float(worst_ratio))


def noise_band(O, pr, sc, i, trials=40, seed=0, settings=None):
    """what double precision leaves open on QP i of `sc`: (the oracle's result, median, max) of the change of the oracle's GRFs over `trials` one-ulp perturbations of one
    word of x0.  On almost every QP this is ~1e-10 N; on the few where a small rho makes the x-update ill-conditioned it reaches 1e-5 N, and there a comparison between
    two double-precision implementations of the same iterate sequence (engine, oracle, OSQP itself) cannot be held to less (tests/tools/outlier_noise_band.py)."""
    st = settings if settings is not None else O.default_settings()
    base = O.mpc_solve(pr, st, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i])
    rng = np.random.default_rng(seed); ds = []
    for _ in range(trials):
        x0 = sc["x0"][i].copy(); j = rng.integers(0, 12); x0[j] = np.nextafter(x0[j], x0[j] + (1.0 if rng.random() < 0.5 else -1.0))
        r = O.mpc_solve(pr, st, x0, sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i])
        ds.append(np.abs(r["grf"].ravel() - base["grf"].ravel()).max())
    return base, float(np.median(ds)), float(max(ds))
