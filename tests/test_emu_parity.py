"""CPU: executes the SAME solver source the GPU runs (csrc/a1mpc_solver.hpp) lane-for-lane on host fibers
(tests/emu, a test double for the DPP primitives) and compares with the oracle -- algorithm, lane mapping, LDS
layout and synchronisation are validated without a GPU.  The real DPP / LDS hardware path is covered by -m gpu."""
import numpy as np
import pytest

import emu
from helpers import TOL_FORCE_BALANCE_N, compare, oracle_batch, oracle_params


@pytest.mark.parametrize("gen,kw,n", [("scenario_T", {}, 1), ("config3_random_flat", dict(nb=8), 6), ("config4_random_h16", dict(nb=4), 2),
                                       ("config5_divergent", dict(nb=4), 2), ("config3_random_flat", dict(nb=4, param_set="hardware"), 3)])
def test_emulated_kernel_matches_oracle_default(oracle, scen, gen, kw, n):
    sc = getattr(scen, gen)(**kw)
    out = emu.solve(sc, n)
    ref = oracle_batch(oracle, sc, n)
    assert (out["nfact"] == ref["nfact"]).all()
    compare(out, ref, tol=1e-8, min_same=1.0)


def test_emulated_kernel_exact_mode(oracle, scen):
    sc = scen.config3_random_flat(nb=2)
    out = emu.solve(sc, 2, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000)
    ref = oracle_batch(oracle, sc, 2, settings=oracle.exact_settings())
    compare(out, ref, tol=1e-8, min_same=1.0)


def test_emulated_warm_start_sequence(oracle, scen):
    sc = scen.config2_trot_sequence(6)
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    wx = np.zeros(120); wy = np.zeros(200); rho = None
    ewx = np.zeros((1, 120)); ewy = np.zeros((1, 200)); erho = np.zeros(1)
    for t in range(6):
        r = oracle.mpc_solve(pr, st, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], warm_x=wx, warm_y=wy, warm_rho=rho)
        wx, wy, rho = r["warm_x"], r["warm_y"], r["rho"]
        one = {k: (sc[k][t:t + 1] if k in ("x0", "xref", "R", "foot", "contact") else sc[k]) for k in sc}
        out = emu.solve(one, 1, warm=(ewx, ewy, erho), warm_start=1)
        assert out["iters"][0] == r["info"].iters, (t, out["iters"], r["info"].iters)
        assert np.abs(out["u"][0] - r["u"]).max() < 1e-8
        assert np.abs(ewy[0] - wy).max() < 1e-6 and abs(erho[0] - rho) < 1e-11 * rho   # rho-estimate: a ratio of nearly cancelled residuals, sensitive to the last bits of P


def test_emulated_balance_qp(oracle, scen):
    sc = scen.balance_random(6)
    out = emu.balance_solve(sc, 6)
    qp, st = oracle.default_qp_params(), oracle.default_settings()
    for b in range(6):
        r = oracle.balance_solve(qp, st, sc["root_acc"][b], sc["R"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
        assert out["iters"][b] == r["info"].iters
        assert np.abs(out["f_world"][b] - r["f_world"]).max() < TOL_FORCE_BALANCE_N


def test_emulated_all_swing_and_max_iter(oracle, scen):
    sc = scen.config3_random_flat(nb=2)
    sc["contact"][:] = 0  # every row an equality: zero forces
    out = emu.solve(sc, 1)
    assert np.abs(out["u"]).max() < 1e-6
    sc = scen.config3_random_flat(nb=2)
    out = emu.solve(sc, 1, max_iter=30, eps_abs=1e-12, eps_rel=1e-12)
    ref = oracle_batch(oracle, sc, 1, settings=oracle.default_settings(max_iter=30, eps_abs=1e-12, eps_rel=1e-12))
    assert out["iters"][0] == 30 and out["status"][0] == ref["status"][0] and np.abs(out["u"] - ref["u"]).max() < 1e-8


@pytest.mark.parametrize("gen,kw,n,rows", [("config3_random_flat", dict(nb=8), 7, 2), ("config4_random_h16", dict(nb=4), 3, 1)])
def test_emulated_split_pipeline(oracle, scen, gen, kw, n, rows):
    """set-up kernel -> prepared state in memory -> persistent ADMM rows pulling QPs from the shared counter"""
    sc = getattr(scen, gen)(**kw)
    out = emu.solve(sc, n, split_rows=rows)
    ref = oracle_batch(oracle, sc, n)
    assert (out["nfact"] == ref["nfact"]).all()
    compare(out, ref, tol=1e-8, min_same=1.0)


def test_emulated_split_pipeline_warm_start(oracle, scen):
    sc = scen.config2_trot_sequence(4)
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    wx = np.zeros(120); wy = np.zeros(200); rho = None
    ewx = np.zeros((1, 120)); ewy = np.zeros((1, 200)); erho = np.zeros(1)
    for t in range(4):
        r = oracle.mpc_solve(pr, st, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], warm_x=wx, warm_y=wy, warm_rho=rho)
        wx, wy, rho = r["warm_x"], r["warm_y"], r["rho"]
        one = {k: (sc[k][t:t + 1] if k in ("x0", "xref", "R", "foot", "contact") else sc[k]) for k in sc}
        out = emu.solve(one, 1, warm=(ewx, ewy, erho), warm_start=1, split_rows=1)
        assert out["iters"][0] == r["info"].iters and np.abs(out["u"][0] - r["u"]).max() < 1e-8


def test_emulated_non_finite_input_returns_zeros_and_status(oracle, scen):
    """a NaN state: the reference would return an uninitialised matrix (S/A1RobotControl.cpp:322,559); here zeros + status"""
    sc = scen.config3_random_flat(nb=2)
    sc["x0"][0, 4] = np.nan
    out = emu.solve(sc, 2)
    ref = oracle_batch(oracle, sc, 2)
    assert out["status"][0] == -7 and (out["grf"][0] == 0).all() and np.isnan(out["u"][0]).all() and np.isnan(ref["u"][0]).all()
    assert out["status"][1] == 1 and np.abs(out["u"][1] - ref["u"][1]).max() < 1e-8


def test_emulated_tick_records_N1(oracle, scen):
    """x0 / x_ref built on the device from the 22-number tick record == the reference's own builder
    (S/A1RobotControl.cpp:452-488, restated by oracle.mpc_reference) followed by the same solve"""
    sc = scen.config3_random_flat(nb=4)
    out = emu.solve_ticks(sc, 4)
    ref = oracle_batch(oracle, sc, 4)
    compare(out, ref, tol=1e-8, min_same=1.0)
    for b in range(4):  # the scenario's x_ref is what the oracle's restatement of :470-488 builds from the same tick
        k = sc["tick"][b]
        xr = oracle.mpc_reference(10, sc["params"]["dt"], k[0:3], k[3:6], sc["R"][b], k[12:15], k[15:18], k[18:21], k[21])
        assert np.abs(xr - sc["xref"][b]).max() < 1e-12


@pytest.mark.parametrize("over", [dict(scaling_iters=0), dict(alpha=1.0, check_every=10, adaptive_rho_every=35), dict(rho0=1.0, adaptive_rho=0)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_emulated_non_default_settings(oracle, scen, over):
    names = dict(scaling_iters="scaling", check_every="check_termination", adaptive_rho_every="adaptive_rho_interval", rho0="rho")
    sc = scen.config3_random_flat(nb=3)
    out = emu.solve(sc, 3, **over)
    ref = oracle_batch(oracle, sc, 3, settings=oracle.default_settings(**{names.get(k, k): v for k, v in over.items()}))
    compare(out, ref, tol=1e-8, min_same=1.0)


def test_emulated_positive_fz_min_first_iteration(oracle, scen):
    sc = scen.config3_random_flat(nb=3)
    sc["params"] = dict(sc["params"], fz_min=5.0)
    out = emu.solve(sc, 3)
    compare(out, oracle_batch(oracle, sc, 3), tol=1e-8, min_same=1.0)
