"""GPU, through the C ABI: OSQP settings other than the defaults, friction / force limits, refused configurations, a1mpc_update_config -- vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import *  # noqa: F401,F403  (_engine, _strided_inputs, tick_inputs, TICK_STATE, _oracle_update_ticks, SETTINGS_CASES)
from gpu_common import _engine, _oracle_update_ticks, _strided_inputs  # noqa: F401
from helpers import TOL_FORCE_BALANCE_N, TOL_FORCE_N, compare, exact_resolver, noise_band, oracle_batch, oracle_params, take  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("over", SETTINGS_CASES, ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_non_default_osqp_settings(pkg, oracle, scen, over):
    """every OSQP setting the ABI exposes, away from its default: same iterates as the oracle run with the same setting"""
    sc = scen.config3_random_flat(nb=48)
    with _engine(pkg, sc, 48, warm_start=0, **over) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    ref = oracle_batch(oracle, sc, settings=oracle.default_settings(**over))
    compare(out, ref, min_same=1.0)


@pytest.mark.parametrize("seed", range(100, 112))
def test_random_setting_combinations(pkg, oracle, scen, seed):
    """several knobs away from their defaults at once (the draws of tests/tools/soak_settings.py, seeds 100-111: OSQP settings x friction /
    force limits x horizon),
    64 QPs each: same iteration count and status on every QP, forces within the bar.  (The soak's 660 combinations are in
    profiles/r03_settings_soak*.txt; the handful
    that exceed the bar are combinations on which the oracle's own two linear-system back ends part by more,
    profiles/r03_settings_soaks_second_round.txt.)"""
    rng = np.random.default_rng(seed)
    H = int(rng.choice([10, 10, 16, 20]))
    over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])),
            rho=float(10 ** rng.uniform(-2, 0.3)),
                sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])),
                        adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])),
                        adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4, 1e-5])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
    over["eps_rel"] = over["eps_abs"]
    gen = {10: scen.config3_random_flat, 16: scen.config4_random_h16, 20: scen.config5_divergent}[H]
    sc = gen(nb=64, seed=7000 + seed)
    sc["params"] = dict(sc["params"], mu=float(rng.choice([0.3, 0.3, 0.6, 0.15])), fz_min=float(rng.choice([0.0, 0.0, 0.0, 5.0])),
            fz_max=float(rng.choice([180.0, 180.0, 120.0, 60.0])))
    with _engine(pkg, sc, 64, warm_start=0, **over) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    compare(out, oracle_batch(oracle, sc, settings=oracle.default_settings(**over)), min_same=1.0)


def test_infeasible_configuration_is_refused_where_osqp_would_report_primal_infeasibility(pkg, oracle, scen):
    """fz_max < 0 puts every stance leg's fz <= fz_max < 0 against a pyramid that asks fz >= 0.  The oracle -- which evaluates OSQP's certificates
    (auxil.c is_primal_infeasible) -- answers PRIMAL_INFEASIBLE (-3) and zero forces for every QP; the engine, which does not evaluate them, refuses the configuration
    with A1MPC_ERR_INVALID_ARGUMENT at a1mpc_create and at a1mpc_update_config (include/a1mpc.h: the statuses -3 / -4 are unreachable on every accepted
    configuration), and a refused update leaves the live handle exactly as it was."""
    sc = scen.config3_random_flat(nb=16)
    bad = dict(sc["params"], fz_min=-10.0, fz_max=-5.0)
    pr = oracle.mpc_params(10, bad["dt"], bad["mu"], bad["fz_min"], bad["fz_max"], bad["q"], bad["r"], bad["mass"], bad["inertia"])
    ref = oracle.mpc_solve_batch(pr, oracle.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    stance = sc["contact"].any(axis=1)
    assert (ref["status"][stance] == -3).all() and not ref["grf"][stance].any()      # what the reference's OSQP would conclude
    with pytest.raises(pkg.A1MpcError, match="fz_max"):
        pkg.Engine(pkg.make_config(bad, 10), 16, 0)
    with _engine(pkg, sc, 16, warm_start=0) as eng:
        a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        for over in (dict(fz_min=-10.0, fz_max=-5.0), dict(fz_min=200.0), dict(mu=-0.3), dict(mass=float("nan"))):
            with pytest.raises(pkg.A1MpcError):
                eng.update_config(pkg.make_config(dict(sc["params"], **over), 10, warm_start=0))
        b = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        assert np.array_equal(a["grf"], b["grf"]) and np.array_equal(a["iters"], b["iters"]) and (b["status"] == 1).all()
        qp = pkg.BalanceConfig(); eng.lib.a1mpc_default_balance_config(C.byref(qp)); qp.F_max = -1.0
        with pytest.raises(pkg.A1MpcError):
            eng.balance_solve(np.zeros((1, 6)), np.eye(3).reshape(1, 9), np.eye(3).reshape(1, 9), np.zeros((1, 12)), np.ones((1, 4), np.uint8), qp=qp)


@pytest.mark.parametrize("mu,fz_min,fz_max", [(0.6, 0.0, 120.0), (0.3, 5.0, 180.0), (0.15, 0.0, 60.0)])
def test_other_friction_and_force_limits(pkg, oracle, scen, mu, fz_min, fz_max):
    """fz_min > 0 excludes u = 0 from the box: OSQP's first iteration (z0 = 0 not projected) needs the dedicated code path"""
    sc = scen.config3_random_flat(nb=48)
    sc["params"] = dict(sc["params"], mu=mu, fz_min=fz_min, fz_max=fz_max)
    with _engine(pkg, sc, 48, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    compare(out, oracle_batch(oracle, sc), min_same=1.0)


def test_update_config_dt_and_warm_start_io(pkg, oracle, scen):
    """dt (the reference passes the measured loop dt when use_sim_time is "true", S/A1RobotControl.cpp:465), weights
    and mass can change on a live handle (a1mpc_update_config); a1mpc_warm_start / a1mpc_get_warm_start move the carried workspace."""
    n = 32
    sc = scen.config3_random_flat(nb=n)
    p2 = dict(sc["params"], dt=0.004, mass=13.0)
    xr2 = sc["xref"].copy()   # x_ref as the caller builds it with the other dt
    tk = sc["tick"]
    xr2 = scen.build_reference(10, 0.004, tk[:, 0:3], tk[:, 3:6], sc["R"].reshape(n, 3, 3), tk[:, 12:15], tk[:, 15:18], tk[:, 18:21],
            tk[:, 21])
    with _engine(pkg, sc, n, warm_start=1) as eng:
        a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        wx, wy, rho = eng.get_warm_start(n)
        eng.update_config(pkg.make_config(p2, 10, warm_start=1))
        b = eng.solve(sc["x0"], xr2, sc["R"], sc["foot"], sc["contact"], want_u=True)
    pr = oracle.mpc_params(10, p2["dt"], p2["mu"], p2["fz_min"], p2["fz_max"], p2["q"], p2["r"], p2["mass"], p2["inertia"])
    st = oracle.default_settings(warm_start=1)
    for i in range(0, n, 4):
        r = oracle.mpc_solve(pr, st, sc["x0"][i], xr2[i], sc["R"][i], sc["foot"][i], sc["contact"][i], warm_x=wx[i], warm_y=wy[i],
                warm_rho=rho[i])
        assert b["iters"][i] == r["info"].iters and np.abs(b["u"][i] - r["u"]).max() <= TOL_FORCE_N, i
    # a workspace written through the ABI is the one the next solve starts from
    with _engine(pkg, sc, n, warm_start=1) as eng:
        eng.set_warm_start(wx, wy, rho)
        eng.update_config(pkg.make_config(p2, 10, warm_start=1))
        c = eng.solve(sc["x0"], xr2, sc["R"], sc["foot"], sc["contact"], want_u=True)
    assert np.array_equal(b["u"], c["u"]) and np.array_equal(b["iters"], c["iters"])
    assert np.isfinite(a["u"]).all()


@pytest.mark.parametrize("seed", [1071, 1217, 1160, 1501])
def test_settings_above_the_parity_bar_are_the_checkers_own_rounding(pkg, oracle, scen, seed):
    """the random-settings soak (tests/tools/soak_settings.py) has a handful of combinations -- scaling 0 / 2, sigma
    ~ 1e-7, rho re-adapted every 10
    iterations, 60-iteration cut-offs -- on which engine and oracle stop at the same iteration with forces 1e-4 ... 1e-1 N apart: ADMM
    amplifies last-bit differences of
    the two linear solves there (the oracle's own two back ends part by more).  Gated here with the x87 extended-precision build of the
    oracle as the yardstick: on the
    three QPs of each such combination with the largest engine-vs-oracle difference all three runs stop at the same iteration, and the
    engine's distance to the
    extended-precision answer is of the order of the double-precision oracle's own (<= 5 x; it is the closer one on most)."""
    import x87
    n = 256
    rng = np.random.default_rng(seed)
    H = int(rng.choice([10, 10, 16, 20]))     # (the draw sequence of tests/tools/soak_settings.py)
    over = dict(scaling=int(rng.choice([0, 2, 10, 10, 15])), alpha=float(rng.choice([1.0, 1.6, 1.6, rng.uniform(1.05, 1.9)])),
            rho=float(10 ** rng.uniform(-2, 0.3)),
                sigma=float(10 ** rng.uniform(-7, -4)), check_termination=int(rng.choice([5, 10, 25, 25, 40])),
                        adaptive_rho=int(rng.choice([0, 1, 1, 1])),
                adaptive_rho_interval=int(rng.choice([0, 10, 25, 35, 50, 100])),
                        adaptive_rho_tolerance=float(rng.choice([1.5, 2.0, 5.0, 5.0])),
                eps_abs=float(rng.choice([1e-3, 1e-3, 1e-4, 1e-5])), max_iter=int(rng.choice([60, 400, 4000, 4000])))
    over["eps_rel"] = over["eps_abs"]
    gen = {10: scen.config3_random_flat, 16: scen.config4_random_h16, 20: scen.config5_divergent}[H]
    sc = gen(nb=n, seed=7000 + seed)
    p = dict(sc["params"], mu=float(rng.choice([0.3, 0.3, 0.6, 0.15])), fz_min=float(rng.choice([0.0, 0.0, 0.0, 5.0])),
            fz_max=float(rng.choice([180.0, 180.0, 120.0, 60.0])))
    with pkg.Engine(pkg.make_config(p, H, warm_start=0, **over), n, 0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    pr = oracle.mpc_params(H, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    ref = oracle.mpc_solve_batch(pr, oracle.default_settings(**over), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    assert (out["iters"] == ref["iters"]).all()
    # (at the iteration limit OSQP's "solved inaccurate" test, 10 x the tolerances, can sit on the same knife edge)
    st_diff = int((out["status"] != ref["status"]).sum())
    assert st_diff <= n // 50, st_diff
    dd = np.abs(out["grf"].reshape(n, 12) - ref["grf"].reshape(n, 12)).max(1)
    xpr = x87.params(p, H); xst = x87.settings(**over)
    rows = []
    for i in np.argsort(-dd)[:3]:
        xr = x87.mpc_solve(xpr, xst, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i])
        d_e = float(np.abs(out["grf"][i] - xr["grf"]).max()); d_o = float(np.abs(ref["grf"][i] - xr["grf"]).max())
        rows.append((int(i), float(dd[i]), d_e, d_o))
        assert xr["iters"] == out["iters"][i], (seed, i, xr["iters"], out["iters"][i])
        assert d_e <= 5.0 * d_o + TOL_FORCE_N, (seed, rows)
    print(f"settings seed {seed} (h = {H}, {over}): worst engine-vs-oracle {dd.max():.2e} N, {st_diff} status differences; "
          f"(qp, engine-vs-oracle, engine-vs-x87, oracle-vs-x87): {rows}")
