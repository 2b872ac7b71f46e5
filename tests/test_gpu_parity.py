"""-m gpu: the HIP path, called through the C ABI (liba1mpc.so), against the oracle on identical inputs."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL_FORCE_BALANCE_N, TOL_FORCE_N, compare, exact_resolver, noise_band, oracle_batch, oracle_params, take

pytestmark = pytest.mark.gpu


def _engine(pkg, sc, max_batch, **osqp):
    cfg = pkg.make_config(sc["params"], sc["horizon"], **osqp)
    return pkg.Engine(cfg, max_batch=max_batch, device=0)


def test_fixture_T_default_and_exact(pkg, oracle, scen):
    """S/test/test_mpc.cpp:18-60 inputs (stand, contacts FL+RL, cold start)."""
    sc = scen.scenario_T()
    with _engine(pkg, sc, 4, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    ref = oracle_batch(oracle, sc)
    compare(out, ref, min_same=1.0)
    with _engine(pkg, sc, 4, warm_start=0, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000) as eng:
        oute = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    refe = oracle_batch(oracle, sc, settings=oracle.exact_settings())
    compare(oute, refe, min_same=1.0)
    f = oute["grf"].reshape(4, 3)
    assert abs(f[0, 1]) == pytest.approx(0.3 * f[0, 2], rel=1e-6)  # friction row active: |fy| = mu fz
    assert np.abs(f[1]).max() < 1e-6 and np.abs(f[3]).max() < 1e-6  # swing legs


@pytest.mark.parametrize("name,gen,n", [("config3_h10", "config3_random_flat", 512), ("config4_h16", "config4_random_h16", 192),
                                         ("config5_h20", "config5_divergent", 192)])
def test_randomized_configs_default_settings(pkg, oracle, scen, name, gen, n):
    sc = getattr(scen, gen)(nb=n)
    with _engine(pkg, sc, n, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    ref = oracle_batch(oracle, sc)
    # (a QP with another iteration count is checked against the exact-mode optimum, not dropped)
    r = compare(out, ref, resolve=exact_resolver(oracle, sc))
    print(name, r, "iters", np.unique(out["iters"], return_counts=True))


def test_parameter_sets(pkg, oracle, scen):
    for ps in ("hardware", "isaac", "ctrl_default"):
        sc = scen.config3_random_flat(nb=64, param_set=ps)
        with _engine(pkg, sc, 64, warm_start=0) as eng:
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        compare(out, oracle_batch(oracle, sc), resolve=exact_resolver(oracle, sc))


def test_exact_mode_h10(pkg, oracle, scen):
    sc = scen.config3_random_flat(nb=64)
    with _engine(pkg, sc, 64, warm_start=0, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    ref = oracle_batch(oracle, sc, settings=oracle.exact_settings())
    # named exception to MIN_SAME_ITERS = 1.0: at eps 1e-10 the termination threshold is inside the round-off of the residuals themselves (the residual of a converged
    # QP is ~1e-11 +- 1e-12 of summation-order noise), so the checkpoint at which it passes may differ by one; the resolver holds such QPs to the exact optimum
    compare(out, ref, min_same=0.9, resolve=exact_resolver(oracle, sc, settings=oracle.exact_settings()))
    # whatever the iteration count, both are the optimum
    assert np.abs(out["u"] - ref["u"]).max() < 1e-5


def test_ragged_and_edge_batches(pkg, oracle, scen):
    """n not a multiple of the 4 QPs per workgroup, n = 1, n = 0, n > max_batch."""
    sc = scen.config3_random_flat(nb=11)
    ref = oracle_batch(oracle, sc)
    with _engine(pkg, sc, 16, warm_start=0) as eng:
        for n in (1, 2, 3, 5, 11):
            s = take(sc, n)
            out = eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"], want_u=True)
            r = {k: v[:n] for k, v in ref.items() if v is not None}
            compare(out, r, min_same=1.0)
        s = take(sc, 0)
        out = eng.solve(np.zeros((0, 13)), np.zeros((0, 130)), np.zeros((0, 9)), np.zeros((0, 12)), np.zeros((0, 4), np.uint8))
        assert out["grf"].shape == (0, 12)
    with _engine(pkg, sc, 4, warm_start=0) as eng:
        with pytest.raises(pkg.A1MpcError):
            eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])


def test_all_contact_patterns(pkg, oracle, scen):
    """every one of the 16 contact patterns incl. 0000 (all rows equalities => zero forces)."""
    sc = scen.config3_random_flat(nb=16)
    sc["contact"] = ((np.arange(16)[:, None] >> np.arange(4)[None, :]) & 1).astype(np.uint8)
    with _engine(pkg, sc, 16, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    compare(out, oracle_batch(oracle, sc), min_same=1.0)
    assert np.abs(out["grf"][0]).max() < 1e-3
    # OSQP-default accuracy on the swing-leg equalities
    assert (np.abs(out["u"].reshape(16, -1, 4, 3)[:, 0][sc["contact"] == 0]) < 0.1).all()


def test_warm_started_tick_sequence(pkg, oracle, scen):
    """config 2: sequential ticks of one robot, warm start + carried rho (S/A1RobotControl.cpp:522-538)."""
    nt = 40
    sc = scen.config2_trot_sequence(nt)
    pr = None
    from helpers import oracle_params
    pr = oracle_params(oracle, sc)
    st = oracle.default_settings(warm_start=1)
    h = sc["horizon"]
    wx = np.zeros(12 * h); wy = np.zeros(20 * h); rho = None
    with _engine(pkg, sc, 1, warm_start=1) as eng:
        for t in range(nt):
            out = eng.solve(sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], want_u=True)
            r = oracle.mpc_solve(pr, st, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], warm_x=wx, warm_y=wy,
                                 warm_rho=rho)
            wx, wy, rho = r["warm_x"], r["warm_y"], r["rho"]
            assert out["iters"][0] == r["info"].iters, (t, out["iters"], r["info"].iters)
            assert np.abs(out["u"][0] - r["u"]).max() < TOL_FORCE_N
        eng.reset_warm_start()
        out = eng.solve(sc["x0"][0], sc["xref"][0], sc["R"][0], sc["foot"][0], sc["contact"][0], want_u=True)
        r0 = oracle.mpc_solve(pr, oracle.default_settings(), sc["x0"][0], sc["xref"][0], sc["R"][0], sc["foot"][0], sc["contact"][0])
        assert out["iters"][0] == r0["info"].iters and np.abs(out["u"][0] - r0["u"]).max() < TOL_FORCE_N


def test_balance_qp(pkg, oracle, scen):
    """compute_grf's balance branch (S/A1RobotControl.cpp:377-444) = the H = 1 member of the kernel family."""
    sc = scen.balance_random(256)
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, 256, 0) as eng:
        out = eng.balance_solve(sc["root_acc"], sc["R"], sc["Rz"], sc["foot"], sc["contact"])
        qp, st = oracle.default_qp_params(), oracle.default_settings()
        worst = 0.0
        for b in range(256):
            r = oracle.balance_solve(qp, st, sc["root_acc"][b], sc["R"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
            assert out["iters"][b] == r["info"].iters and out["status"][b] == r["info"].status
            worst = max(worst, np.abs(out["f_world"][b] - r["f_world"]).max(), np.abs(out["grf"][b] - r["grf"]).max())
        print("balance QP: max |df| vs oracle over 256 QPs =", worst)
        assert worst < TOL_FORCE_BALANCE_N
        s1 = scen.config1_balance_stand()
        o1 = eng.balance_solve(s1["root_acc"], s1["R"], s1["Rz"], s1["foot"], s1["contact"])
        assert np.allclose(o1["grf"].reshape(4, 3)[:, 2], 12.0 * 9.8 / 4, atol=0.02)  # ~ m g / 4 per leg


def test_size_independent_properties_full_batch(pkg, scen):
    """BASELINE config 3 at its full size (4096, h=10): properties that need no oracle."""
    sc = scen.config3_random_flat()
    n = len(sc["x0"])
    with _engine(pkg, sc, n, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        # batch-order invariance: a permuted batch gives the permuted answer bit for bit
        perm = np.random.default_rng(0).permutation(n)
        outp = eng.solve(sc["x0"][perm], sc["xref"][perm], sc["R"][perm], sc["foot"][perm], sc["contact"][perm], want_u=True)
    assert (outp["u"] == out["u"][perm]).all() and (outp["iters"] == out["iters"][perm]).all()
    assert (out["status"] == 1).all()
    u = out["u"].reshape(n, 10, 4, 3)
    mu, tol = 0.3, 0.5  # OSQP's default 1e-3 tolerances leave O(0.1 N) constraint violation
    assert (u[..., 2] >= -tol).all() and (u[..., 2] <= 180 + tol).all()
    assert (np.abs(u[..., 0]) <= mu * u[..., 2] + tol).all() and (np.abs(u[..., 1]) <= mu * u[..., 2] + tol).all()
    assert np.abs(u.transpose(0, 2, 1, 3)[sc["contact"] == 0]).max() < tol  # swing legs carry no force
    # first-step GRF = R' u_0
    R = sc["R"].reshape(n, 3, 3)
    g = np.einsum("nji,nlj->nli", R, u[:, 0])
    assert np.abs(g.reshape(n, 12) - out["grf"]).max() < 1e-9


@pytest.mark.parametrize("gen,n,history", [("config3_random_flat", 200, True),    # latency kernel (the rows of a wave share a set-up)
                                           ("config3_random_flat", 1500, True),   # fused kernel, two QPs per wave
                                           ("config3_random_flat", 3000, True),   # split pipeline, longest-first queue
                                           ("config3_random_flat", 3000, False),  # split pipeline, index-order queue
                                           # fused and latency kernels' quads of rows
                                           ("config4_random_h16", 700, True), ("config5_divergent", 300, True),
                                           # persistent quads (CU-wide at h = 16) vs fused quads
                                           ("config4_random_h16", 1500, True), ("config5_divergent", 1300, True)])
def test_result_does_not_depend_on_position_or_history(pkg, scen, gen, n, history):
    """Every kernel path: a QP's result is bit for bit the same wherever it sits in the batch, whatever its wave-mates are and
    whatever the row solved before (another batch in between).  A violated DPP read hazard or a stale LDS word shows up here."""
    sc = getattr(scen, gen)(nb=n)
    args = lambda idx: (sc["x0"][idx], sc["xref"][idx], sc["R"][idx], sc["foot"][idx], sc["contact"][idx])
    rng = np.random.default_rng(11)
    with _engine(pkg, sc, n, warm_start=0) as eng:
        eng.set_schedule(history)
        out = eng.solve(*args(np.arange(n)), want_u=True)
        perm = rng.permutation(n)
        outp = eng.solve(*args(perm), want_u=True)
        sub = np.sort(rng.choice(n, n // 3, replace=False))  # a different batch size: other wave-mates, other queue order
        outs = eng.solve(*args(sub), want_u=True)
        again = eng.solve(*args(np.arange(n)), want_u=True)
    for o, idx in ((outp, perm), (outs, sub), (again, np.arange(n))):
        assert (o["u"] == out["u"][idx]).all() and (o["iters"] == out["iters"][idx]).all() and (o["status"] == out["status"][idx]).all()


def test_non_finite_input_returns_zeros_and_status(pkg, oracle, scen):
    """a NaN state must not poison its neighbours in the wave: zeros + status -7 for it, exact answers for the others"""
    for n in (8, 512):  # fused kernel, split pipeline
        sc = scen.config3_random_flat(nb=n)
        sc["x0"][1, 4] = np.nan; sc["foot"][5, 2] = np.inf
        with _engine(pkg, sc, n, warm_start=0) as eng:
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        bad = np.zeros(n, bool); bad[[1, 5]] = True
        assert (out["status"][bad] == -7).all() and (out["grf"][bad] == 0).all()
        ref = oracle_batch(oracle, take(sc, 16))
        good = ~bad[:16]
        assert (out["iters"][:16][good] == ref["iters"][good]).all()
        assert np.abs(out["u"][:16][good] - ref["u"][good]).max() < TOL_FORCE_N


@pytest.mark.parametrize("gen,n,h", [("config4_random_h16", 8192, 16), ("config5_divergent", 32768, 20)])
def test_full_size_configs_4_and_5_properties(pkg, oracle, scen, gen, n, h):
    """BASELINE configs[3] (65536 x h16 over 8 GPUs = 8192 per GPU) and configs[4] (32768 x h20) at full per-GPU size: the oracle
    checks a strided sample, the whole batch is checked through size-independent properties."""
    sc = getattr(scen, gen)(nb=n)
    with _engine(pkg, sc, n, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    assert (out["status"] == 1).all()
    idx = np.arange(0, n, n // 48)
    sub = {k: (sc[k][idx] if k in ("x0", "xref", "R", "foot", "contact") else sc[k]) for k in sc}
    ref = oracle_batch(oracle, sub)
    compare({k: (v[idx] if v is not None else None) for k, v in out.items()}, ref, min_same=1.0)
    u = out["u"].reshape(n, h, 4, 3)
    mu, tol = 0.3, 1.0  # OSQP-default accuracy
    assert (u[..., 2] >= -tol).all() and (u[..., 2] <= 180 + tol).all()
    assert (np.abs(u[..., 0]) <= mu * u[..., 2] + tol).all() and (np.abs(u[..., 1]) <= mu * u[..., 2] + tol).all()
    assert np.abs(u.transpose(0, 2, 1, 3)[sc["contact"] == 0]).max() < tol
    R = sc["R"].reshape(n, 3, 3)
    assert np.abs(np.einsum("nji,nlj->nli", R, u[:, 0]).reshape(n, 12) - out["grf"]).max() < 1e-9


def test_tick_records_N1(pkg, oracle, scen):
    """SURVEY 8(f) N1 through the C ABI: a1mpc_solve_batch_ticks == a1mpc_solve_batch on the x0 / x_ref the reference would build"""
    for n in (64, 1024):
        sc = scen.config3_random_flat(nb=n)
        with _engine(pkg, sc, n, warm_start=0) as eng:
            a = eng.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"], want_u=True)
            b = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        assert (a["iters"] == b["iters"]).all() and np.abs(a["u"] - b["u"]).max() < TOL_FORCE_N
        compare({k: v[:32] if v is not None else None for k, v in a.items()}, oracle_batch(oracle, take(sc, 32)), min_same=1.0)


def test_device_pointer_entry_and_batched_warm_start(pkg, oracle, scen):
    """a1mpc_solve_batch_device (asynchronous, caller's stream, torch tensors in HBM) == the host-pointer entry; and a batch of
    300 robots ticked twice with warm start matches 300 sequentially warm-started oracle solvers"""
    import torch
    n = 300
    sc = scen.config3_random_flat(nb=n)
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(sc[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")}
    grf = torch.zeros((n, 12), dtype=torch.float64, device=dev); u = torch.zeros((n, 120), dtype=torch.float64, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev); stt = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    with _engine(pkg, sc, n, warm_start=1) as eng:
        host1 = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)      # tick 1 (cold: zeros)
        eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, u, it, stt, stream=st.cuda_stream)  # tick 2 (warm)
        st.synchronize()
    from helpers import oracle_params
    pr = oracle_params(oracle, sc); so = oracle.default_settings(warm_start=1)
    for b in range(0, n, 7):
        r1 = oracle.mpc_solve(pr, so, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], warm_x=np.zeros(120),
                warm_y=np.zeros(200))
        assert host1["iters"][b] == r1["info"].iters and np.abs(host1["u"][b] - r1["u"]).max() < TOL_FORCE_N
        r2 = oracle.mpc_solve(pr, so, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], warm_x=r1["warm_x"],
                warm_y=r1["warm_y"],
                              warm_rho=r1["rho"])
        assert int(it[b]) == r2["info"].iters, (b, int(it[b]), r2["info"].iters)
        assert np.abs(u[b].cpu().numpy() - r2["u"]).max() < TOL_FORCE_N


SETTINGS_CASES = [dict(scaling=0), dict(scaling=3), dict(alpha=1.0), dict(alpha=1.8), dict(rho=1.0), dict(rho=0.01, adaptive_rho=0),
                  dict(check_termination=10), dict(adaptive_rho_interval=50), dict(check_termination=10, adaptive_rho_interval=35),
                  dict(max_iter=30), dict(sigma=1e-4), dict(adaptive_rho_interval=0), dict(adaptive_rho_interval=0, check_termination=10),
                          dict(eps_abs=1e-5, eps_rel=1e-5), dict(adaptive_rho_tolerance=2.0)]


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


def test_horizon_1_mpc(pkg, oracle, scen):
    """the MPC path at horizon 1 (n = 12, m = 20), the shape BASELINE's north star pairs with the balance QP"""
    sc = scen.config3_random_flat(nb=64, horizon=1)
    with _engine(pkg, sc, 64, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    compare(out, oracle_batch(oracle, sc), min_same=1.0)


def test_update_plan_N2a_bit_exact(pkg, oracle, scen):
    """gait counters, planned contacts, Raibert foothold (S/A1RobotControl.cpp:148-202) -- element-wise arithmetic,
    so the bar is BIT-exact against the oracle's restatement (both built without FMA contraction)."""
    rng = np.random.default_rng(7)
    n = 5000
    yaw = rng.uniform(-np.pi, np.pi, n); roll = rng.uniform(-0.2, 0.2, n); pit = rng.uniform(-0.2, 0.2, n)
    R = scen.rot_zyx(roll, pit, yaw).reshape(n, 9); Rz = scen.rot_zyx(0 * yaw, 0 * yaw, yaw).reshape(n, 9)
    mm = (rng.random(n) < 0.8).astype(np.uint8)
    gc = rng.uniform(0, 240, (n, 4)); gc[::7] = [0, 120, 120, 0]; gc[::11, 0] = 239.0  # wrap-around through fmod
    spd = rng.choice([1.0, 1.5, 2.0, 3.0], size=(n, 4))
    v = rng.normal(0, 0.6, (n, 3)); vd = rng.normal(0, 0.6, (n, 3)); vd[::5] *= 10  # saturates FOOT_DELTA_*_LIMIT
    pos = rng.normal(0, 2.0, (n, 3))
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.update_plan(mm, gc, spd, v, Rz, R, pos, vd)
    gp = oracle.gait_params([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35])
    for b in range(0, n, 3):
        g2, pc, rel, ab, wo = oracle.update_plan(gp, mm[b], gc[b], spd[b], v[b], Rz[b], R[b], pos[b], vd[b])
        assert (out["gait_counter"][b] == g2).all() and (out["plan_contacts"][b] == pc).all()
        assert (out["foot_pos_target_rel"][b] == rel).all() and (out["foot_pos_target_abs"][b] == ab).all()
        assert (out["foot_pos_target_world"][b] == wo).all()
    assert (out["plan_contacts"][mm == 0] == 1).all() and (np.abs(out["foot_pos_target_rel"].reshape(n, 4, 3)[:, :, 0] - [0.17, 0.17,
            -0.17, -0.17]) <= 0.1 + 1e-15).all()


def test_joint_torques_N3_bit_exact(pkg, oracle, scen):
    """tau = J'(-f) on stance legs, J^-1 (km .* f_kin) by partial-pivot LU on swing legs, gravity term, NaN guard
    (S/A1RobotControl.cpp:289-319): bit-exact against the oracle's restatement."""
    rng = np.random.default_rng(11)
    n = 4000
    Jb = rng.normal(0, 0.2, (n, 4, 9)); Jb[:, :, [0, 4, 8]] += rng.choice([-0.3, 0.3], size=(n, 4, 3))  # every pivot pattern occurs
    Jb[5, 1] = 0.0  # singular block -> NaN -> previous torque kept
    c = (rng.random((n, 4)) < 0.5).astype(np.uint8); act = (rng.random(n) < 0.9).astype(np.uint8)
    grf = rng.normal(0, 40, (n, 12)); fk = rng.normal(0, 20, (n, 12)); tg = rng.normal(0, 1, (n, 12)); prev = rng.normal(0, 5, (n, 12))
    km = np.array([0.1, 0.1, 0.04])
    c[5, 1] = 0; act[5] = 1
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, n, 0) as eng:
        tau = eng.joint_torques(act, c, Jb.reshape(n, 36), grf, fk, km, tg, prev)
    for b in range(0, n, 2):
        ref = oracle.joint_torques(act[b], c[b], Jb[b].reshape(36), grf[b], fk[b], km, tg[b], prev[b])
        assert (tau[b] == ref).all(), b
    ref5 = oracle.joint_torques(1, c[5], Jb[5].reshape(36), grf[5], fk[5], km, tg[5], prev[5])
    assert (tau[act == 0] == 0).all() and np.array_equal(tau[5], ref5)
    # 0*inf = NaN is guarded (:314-317), the infinity is not -- like the reference
    assert (tau[5, 3:5] == prev[5, 3:5]).all() and np.isinf(tau[5, 5])


def test_queue_order_does_not_change_results(pkg, scen):
    """a1mpc_set_schedule: longest-first by the previous solve's cost vs index order -- same QPs, bit-identical results, any order"""
    sc = scen.config3_random_flat(nb=3000)
    cfg = pkg.make_config(sc["params"], 10, warm_start=0)
    args = (sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    with pkg.Engine(cfg, 3000, 0) as eng:
        first = eng.solve(*args)            # no history yet: index order
        again = eng.solve(*args)            # ordered by the first solve's costs
        eng.set_schedule(False)
        plain = eng.solve(*args)
    for k in ("grf", "iters", "status"):
        assert np.array_equal(first[k], again[k]) and np.array_equal(first[k], plain[k]), k


def test_random_batches_statistics(pkg, oracle, scen):
    """Arbitrary random batches (seeds that no other test uses): identical iteration counts and statuses, and the stated distribution of
    ||u_gpu - u_oracle||_inf -- median at rounding level, 99.9 % below 1e-7 N, every QP below 1e-5 N.  These batches contain the QPs whose
    rho
    estimate is taken at rho = RHO_MIN, where the dual residual must be carried through the x-update identity."""
    from helpers import TOL_FORCE_ANY_BATCH_N
    worst = 0.0
    for seed in (1002, 1003, 1007, 2024):
        n = 2048
        sc = scen.config3_random_flat(nb=n, seed=seed); p = sc["params"]
        cfg = pkg.make_config(p, 10, warm_start=0)
        with pkg.Engine(cfg, n, 0) as eng:
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        pr = oracle.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
        ref = oracle.mpc_solve_batch(pr, oracle.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        d = np.abs(out["grf"].reshape(n, 12) - ref["grf"].reshape(n, 12)).max(1)
        assert (out["iters"].ravel() == ref["iters"].ravel()).mean() >= 0.999 and (out["status"].ravel() == ref["status"].ravel()).all()
        assert np.median(d) < 1e-10 and np.percentile(d, 99.9) < 1e-7 and d.max() < TOL_FORCE_ANY_BATCH_N, (seed, np.median(d), d.max())
        worst = max(worst, d.max())
    assert worst < TOL_FORCE_ANY_BATCH_N


def test_contact_terrain_N2b_sequence(pkg, oracle, scen):
    """150 ticks of contact logic + moving-window filters + plane fit + terrain pitch for 300 robots, device-resident
    filter state vs the oracle's per-robot state (S/A1RobotControl.cpp:256-282, 566-582, 335-376).  Contacts and the filtered contact
    positions are bit-exact (same arithmetic, no contraction); the terrain angle goes through acos (device math library vs glibc)."""
    rng = np.random.default_rng(21)
    n, ticks = 300, 150
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    states = [oracle.contact_state() for _ in range(n)]
    pitch_g = np.zeros(n); pitch_o = np.zeros(n)
    base = np.outer([0.2, 0.2, -0.2, -0.2], [1.0, 0.0, 0.3]).reshape(12) + np.outer([1, -1, 1, -1], [0.0, 0.13, 0.0]).reshape(12)
    gcs = rng.uniform(0, 240, (n, 4))
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            gcs = np.fmod(gcs + 2.0, 240.0)
            plan = (gcs <= 120).astype(np.uint8)
            ff = rng.uniform(0, 80, (n, 4))
            foot = base + rng.normal(0, 0.03, (n, 12)) + np.tile([0.0, 0.0, -0.3], 4)
            z = np.where(rng.random(n) < 0.9, 0.3, 0.05)
            out = eng.contact_terrain(gcs, plan, ff, foot, z, pitch_g)
            pitch_g = out["root_euler_d_pitch"]
            for b in range(0, n, 7):
                ct, rec, ang, pitch_o[b] = oracle.contact_terrain_step(states[b], gcs[b], plan[b], ff[b], foot[b], z[b], pitch_o[b])
                assert (out["contacts"][b] == ct).all() and (out["foot_pos_recent_contact"][b] == rec).all(), (t, b)
                assert abs(out["terrain_angle"][b] - ang) <= 1e-13 and abs(pitch_g[b] - pitch_o[b]) <= 1e-13, (t, b)
        eng.reset_contact_state()
        out = eng.contact_terrain(gcs, plan, ff, foot, z, np.zeros(n))
        fresh = oracle.contact_state()
        ct, rec, ang, _ = oracle.contact_terrain_step(fresh, gcs[0], plan[0], ff[0], foot[0], z[0], 0.0)
        assert (out["foot_pos_recent_contact"][0] == rec).all() and abs(out["terrain_angle"][0] - ang) <= 1e-13


def test_contact_terrain_N2b_partial_wavefront_and_spare_capacity(pkg, oracle, scen):
    """N2b with n = 70 robots on a handle created for 200: the second wavefront of the launch stages 6 records (the record copy is cut at
    n, the ring regions start
    behind max_batch records) -- every robot against the oracle over a leg-window wrap."""
    rng = np.random.default_rng(77)
    n, cap, ticks = 70, 200, 90
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    states = [oracle.contact_state() for _ in range(n)]
    pitch_g = np.zeros(n); pitch_o = np.zeros(n)
    gcs = rng.uniform(0, 240, (n, 4))
    with pkg.Engine(cfg, cap, 0) as eng:
        for t in range(ticks):
            gcs = np.fmod(gcs + 2.0, 240.0)
            plan = (gcs <= 150).astype(np.uint8); ff = rng.uniform(0, 80, (n, 4)); foot = rng.normal(0, 0.2, (n, 12)); z = np.full(n, 0.3)
            out = eng.contact_terrain(gcs, plan, ff, foot, z, pitch_g); pitch_g = out["root_euler_d_pitch"]
            for b in range(n):
                ct, rec, ang, pitch_o[b] = oracle.contact_terrain_step(states[b], gcs[b], plan[b], ff[b], foot[b], z[b], pitch_o[b])
                assert (out["contacts"][b] == ct).all() and (out["foot_pos_recent_contact"][b] == rec).all(), (t, b)
                assert abs(out["terrain_angle"][b] - ang) <= 1e-13 and abs(pitch_g[b] - pitch_o[b]) <= 1e-13, (t, b)


def test_swing_legs_N4a_sequence(pkg, oracle, scen):
    """swing-leg Bezier targets + foot PD force over 60 ticks for 500 robots (S/A1RobotControl.cpp:204-254).  The carried
    state and foot_pos_cur are bit-exact; the curve uses products for the integer powers (std::pow in the reference): a few ulp."""
    rng = np.random.default_rng(31)
    n, ticks = 500, 60
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    st_g = [np.zeros((n, 12)) for _ in range(3)]; st_o = [np.zeros((n, 12)) for _ in range(3)]
    gcs = rng.uniform(0, 240, (n, 4))
    base = np.array([0.17, 0.15, -0.3, 0.17, -0.15, -0.3, -0.17, 0.15, -0.3, -0.17, -0.15, -0.3])
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            gcs = np.fmod(gcs + 2.0, 240.0)
            yaw = rng.uniform(-3, 3, n); Rz = scen.rot_zyx(0 * yaw, 0 * yaw, yaw).reshape(n, 9)
            foot = base + rng.normal(0, 0.03, (n, 12)); tgt = base + rng.normal(0, 0.05, (n, 12))
            cur, kin = eng.swing_legs(Rz, foot, gcs, tgt, *st_g)
            for b in range(0, n, 11):
                c_o, k_o = oracle.swing_legs(Rz[b], foot[b], gcs[b], tgt[b], st_o[0][b], st_o[1][b], st_o[2][b])
                assert (cur[b] == c_o).all() and (st_g[0][b] == st_o[0][b]).all() and (st_g[1][b] == st_o[1][b]).all(), (t, b)
                assert np.abs(st_g[2][b] - st_o[2][b]).max() <= 1e-15 and np.abs(kin[b] - k_o).max() <= 1e-9, (t, b,
                        np.abs(kin[b] - k_o).max())
                st_o[2][b] = st_g[2][b]  # keep the two state copies from drifting apart by the curve's ulp differences


def test_control_tick_chain(pkg, oracle, scen):
    """The caller-side rows composed the way the reference's 400 Hz loop composes them (S/A1RobotControl.cpp: update_plan ->
    generate_swing_legs_ctrl -> compute_grf [terrain pitch -> MPC] -> compute_joint_torques), 48 robots x 16 ticks with synthetic
    sensor inputs, device entry points vs the oracle functions chained the same way: the joint torques at the end of every tick."""
    rng = np.random.default_rng(77)
    n, ticks, h = 48, 16, 10
    P = scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS
    cfg = pkg.make_config(P, h, warm_start=0)
    pr = oracle.mpc_params(h, P["dt"], P["mu"], P["fz_min"], P["fz_max"], P["q"], P["r"], P["mass"],
            P["inertia"]); st = oracle.default_settings()
    dfp = np.array([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35]); gp = oracle.gait_params(dfp)
    km = np.array([0.1, 0.1, 0.04])
    G = dict(gc=np.tile([0.0, 120.0, 120.0, 0.0], (n, 1)), start=np.zeros((n, 12)), rl=np.tile(dfp, (n, 1)), tl=np.tile(dfp, (n, 1)),
            pitch=np.zeros(n), tau=np.zeros((n, 12)))
    O = dict(gc=G["gc"].copy(), start=np.zeros((n, 12)), rl=G["rl"].copy(), tl=G["tl"].copy(), pitch=np.zeros(n), tau=np.zeros((n, 12)),
             ct=[oracle.contact_state() for _ in range(n)])
    worst = 0.0
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            # synthetic sensors of this tick
            eul = rng.normal(0, 0.05, (n, 3)); eul[:, 2] = rng.uniform(-1, 1, n); pos = np.c_[rng.normal(0, 1, (n, 2)),
                    0.3 + rng.normal(0, 0.01, n)]
            w = rng.normal(0, 0.3, (n, 3)); v = rng.normal(0, 0.3, (n, 3)); vd = np.c_[rng.uniform(-0.5, 0.5, (n, 2)),
                    np.zeros(n)]; wd = np.c_[np.zeros((n, 2)), rng.uniform(-0.5, 0.5, n)]
            R = scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9); Rz = scen.rot_zyx(0 * eul[:, 0], 0 * eul[:, 0],
                    eul[:, 2]).reshape(n, 9)
            foot_rel = dfp + rng.normal(0, 0.02, (n, 12))
            foot_abs = np.einsum("nij,nlj->nli", R.reshape(n, 3, 3), foot_rel.reshape(n, 4, 3)).reshape(n, 12)
            ff = rng.uniform(0, 80, (n, 4)); Jb = rng.normal(0, 0.2, (n, 36)); Jb[:, [0, 4, 8, 9, 13, 17, 18, 22, 26, 27, 31, 35]] += 0.3
            tg = rng.normal(0, 0.5, (n, 12)); mm = np.ones(n, np.uint8); spd = np.full((n, 4), 2.0)
            # ---- device chain
            up = eng.update_plan(mm, G["gc"], spd, v, Rz, R, pos, vd); G["gc"] = up["gait_counter"]
            cur, kin = eng.swing_legs(Rz, foot_abs, G["gc"], up["foot_pos_target_rel"], G["start"], G["rl"], G["tl"])
            ctr = eng.contact_terrain(G["gc"], up["plan_contacts"], ff, foot_abs, pos[:, 2],
                    G["pitch"]); G["pitch"] = ctr["root_euler_d_pitch"]
            eul_d = np.c_[np.zeros(n), G["pitch"], eul[:, 2]]
            tick = scen.pack_tick(eul, pos, w, v, eul_d, vd, wd, np.full(n, 0.3))
            sol = eng.solve_ticks(tick, R, foot_abs, ctr["contacts"])
            G["tau"] = eng.joint_torques(np.ones(n, np.uint8), ctr["contacts"], Jb, sol["grf"], kin, km, tg, G["tau"])
            # ---- oracle chain
            for b in range(n):
                gc2, pc, rel, ab, wo = oracle.update_plan(gp, 1, O["gc"][b], spd[b], v[b], Rz[b], R[b], pos[b], vd[b]); O["gc"][b] = gc2
                c_o, k_o = oracle.swing_legs(Rz[b], foot_abs[b], gc2, rel, O["start"][b], O["rl"][b], O["tl"][b])
                ct, rec, ang, O["pitch"][b] = oracle.contact_terrain_step(O["ct"][b], gc2, pc, ff[b], foot_abs[b], pos[b, 2], O["pitch"][b])
                ed = np.array([0.0, O["pitch"][b], eul[b, 2]])
                xref = oracle.mpc_reference(h, P["dt"], eul[b], pos[b], R[b], ed, vd[b], wd[b], 0.3)
                x0 = scen.pack_x0(eul[b:b + 1], pos[b:b + 1], w[b:b + 1], v[b:b + 1])[0]
                grf = oracle.mpc_solve(pr, st, x0, xref, R[b], foot_abs[b], ct)["grf"]
                O["tau"][b] = oracle.joint_torques(1, ct, Jb[b], grf, k_o, km, tg[b], O["tau"][b])
                assert (ctr["contacts"][b] == ct).all() and (G["gc"][b] == gc2).all()
            worst = max(worst, np.abs(G["tau"] - O["tau"]).max())
            O["tl"][:] = G["tl"]  # the curve's ulp differences must not accumulate into the comparison (see test_swing_legs_N4a_sequence)
    assert worst < 1e-5, worst


def test_leg_state_N4b(pkg, oracle, scen):
    """leg forward kinematics, Jacobians and the frame chain of the joint-state callback (S/GazeboA1ROS.cpp:264-279) for
    3000 robots vs the oracle; sin / cos come from the device math library, so the bar is a few ulp of the 0.4 m leg (1e-14)."""
    rng = np.random.default_rng(41)
    n = 3000
    q = rng.uniform(-1.2, 1.2, (n, 12)); qd = rng.normal(0, 3, (n, 12)); opt = rng.normal(0, 0.01, (4, 3))
    eul = rng.uniform(-0.5, 0.5, (n, 3)); eul[:, 2] = rng.uniform(-3, 3, n); R = scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9)
    pos = rng.normal(0, 2, (n, 3)); vel = rng.normal(0, 1, (n, 3))
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.leg_state(q, qd, R, pos, vel, rho_opt=opt)
    for b in range(0, n, 5):
        ref = oracle.leg_state(q[b], qd[b], R[b], pos[b], vel[b], rho_opt=opt)
        for k, tol in (("foot_pos_rel", 1e-14), ("Jb", 1e-14), ("foot_vel_rel", 1e-13), ("foot_pos_abs", 1e-14), ("foot_vel_abs", 1e-13),
                       ("foot_pos_world", 1e-14), ("foot_vel_world", 1e-13)):
            assert np.abs(out[k][b] - ref[k]).max() <= tol, (b, k, np.abs(out[k][b] - ref[k]).max())


def test_ekf_N4c_sequence(pkg, oracle, scen):
    """A1BasicEKF for 200 robots over 80 ticks, device-resident filter state vs the oracle's dense restatement
    (S/A1BasicEKF.cpp:54-163).  Two checks (the oracle must not move with the kernel): (i) against the PINNED restatement --
    multiply + add, what
    tests/test_ref_pin.py holds to the reference's compiled source -- within 1e-11: the kernel accumulates its four dense products by FMA,
    a rounding per term
    of 18- and 28-term dot products on a contracting filter (the CPU suite measures 1.1e-13 between the two arithmetics over 200 ticks);
    (ii) bit for bit
    against the oracle's FMA variant of the same products (same operation order: what pins the kernel's lane map and elimination)."""
    rng = np.random.default_rng(51)
    n, ticks = 200, 80
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    states = [oracle.ekf_state() for _ in range(n)]; pinned = [oracle.ekf_state() for _ in range(n)]
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            mm = np.where(rng.random(n) < 0.8, 1, 0).astype(np.uint8) if t > 3 else np.zeros(n, np.uint8)
            yaw = rng.uniform(-3, 3, n); eul = rng.normal(0, 0.05, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9)
            fk = base + rng.normal(0, 0.01, (n, 12)); fv = rng.normal(0, 0.3, (n, 12)); acc = np.array([0.0, 0.0, 9.81]) + rng.normal(0,
                    0.3, (n, 3))
            w = rng.normal(0, 0.3, (n, 3)); ff = rng.uniform(0, 160, (n, 4))
            pos, vel, ec = eng.ekf_update(0.0025, mm, ff, R, acc, w, fk, fv)
            for b in range(0, n, 3):
                p_o, v_o, e_o = oracle.ekf_step(states[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], fma=True)
                assert np.array_equal(pos[b], p_o) and np.array_equal(vel[b], v_o) and (ec[b] == e_o).all(), (t, b, pos[b] - p_o,
                        vel[b] - v_o)
                p_p, v_p, e_p = oracle.ekf_step(pinned[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b])
                assert max(np.abs(pos[b] - p_p).max(), np.abs(vel[b] - v_p).max()) <= 1e-11 and (ec[b] == e_p).all(), (t, b, pos[b] - p_p,
                        vel[b] - v_p)


def test_ekf_large_batch_residency(pkg, oracle, scen):
    """Batches of 16 384 robots and more run the EKF in its three-waves-per-SIMD residency (680 instead of 1280 LDS words per robot, the
    same arithmetic in the same order).  16 385 robots (the last workgroup half empty), five ticks from the first-call initialisation on:
    a spread of robots incl. the first and the last against the oracle's FMA variant bit for bit, and the robots of a 200-robot engine
    (the other residency) against the same rows of the large batch bit for bit."""
    rng = np.random.default_rng(52)
    n, ticks, small = 16385, 5, 200
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    sample = sorted({0, 1, 2, small - 1, n // 2, n - 2, n - 1} | set(rng.integers(0, n, 40).tolist()))
    states = {b: oracle.ekf_state() for b in sample}
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    with pkg.Engine(cfg, n, 0) as eng, pkg.Engine(cfg, small, 0) as eng_s:
        for t in range(ticks):
            mm = np.where(rng.random(n) < 0.8, 1, 0).astype(np.uint8) if t > 1 else np.zeros(n, np.uint8)
            yaw = rng.uniform(-3, 3, n); eul = rng.normal(0, 0.05, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9)
            fk = base + rng.normal(0, 0.01, (n, 12)); fv = rng.normal(0, 0.3, (n, 12))
            acc = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.3, (n, 3))
            w = rng.normal(0, 0.3, (n, 3)); ff = rng.uniform(0, 160, (n, 4))
            pos, vel, ec = eng.ekf_update(0.0025, mm, ff, R, acc, w, fk, fv)
            s = slice(0, small)
            pos_s, vel_s, ec_s = eng_s.ekf_update(0.0025, mm[s], ff[s], R[s], acc[s], w[s], fk[s], fv[s])
            assert np.array_equal(pos[s], pos_s) and np.array_equal(vel[s], vel_s) and np.array_equal(ec[s], ec_s), t
            for b in sample:
                p_o, v_o, e_o = oracle.ekf_step(states[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], fma=True)
                assert np.array_equal(pos[b], p_o) and np.array_equal(vel[b], v_o) and (ec[b] == e_o).all(), (t, b, pos[b] - p_o, vel[b] - v_o)


def test_device_pointer_tick_matches_host_pointer_tick(pkg, scen):
    """The *_device variants of the caller-side entry points chained on the GPU (torch tensors, one stream, no host copies between the
    stages) give bit for bit what the host-pointer entries give: leg state -> EKF -> plan -> swing legs -> contacts / terrain -> MPC
    (tick records) -> joint torques, three ticks, 256 robots."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(99)
    n, h = 256, 10
    P = scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS
    cfg = pkg.make_config(P, h, warm_start=0)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    km = np.array([0.1, 0.1, 0.04]); kp = np.array([300.0, 400.0, 400.0]); kd = np.array([8.0, 8.0,
            8.0]); dp_ = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    with pkg.Engine(cfg, n, 0) as eh, pkg.Engine(cfg, n, 0) as ed:
        gait = pkg.engine.GaitConfig(); ed.lib.a1mpc_default_gait_config(C.byref(gait)); ccfg = pkg.engine.ContactConfig(
                ); ed.lib.a1mpc_default_contact_config(C.byref(ccfg))
        fix = np.ascontiguousarray(eh.A1_RHO_FIX); opt = np.zeros((4, 3))
        st_h = dict(gc=np.tile([0.0, 120.0, 120.0, 0.0], (n, 1)), start=np.zeros((n, 12)), rl=np.zeros((n, 12)), tl=np.zeros((n, 12)),
                pitch=np.zeros(n), tau=np.zeros((n, 12)))
        st_d = {k: T(v) for k, v in st_h.items()}
        st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
        for t in range(3):
            q = rng.uniform(-0.8, 0.8, (n, 12)); qd = rng.normal(0, 1, (n, 12)); eul = rng.normal(0, 0.05, (n, 3)); eul[:,
                    2] = rng.uniform(-1, 1, n)
            R = scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9); Rz = scen.rot_zyx(0 * eul[:, 0], 0 * eul[:, 0],
                    eul[:, 2]).reshape(n, 9)
            acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.2, (n, 3)); w = rng.normal(0, 0.2, (n, 3)); ff = rng.uniform(0, 120,
                    (n, 4)); mm = np.ones(n, np.uint8)
            vd = np.c_[rng.uniform(-0.4, 0.4, (n, 2)), np.zeros(n)]; wd = np.c_[np.zeros((n, 2)),
                    rng.uniform(-0.4, 0.4, n)]; spd = np.full((n, 4), 2.0); tg = rng.normal(0, 0.5, (n, 12))
            act = np.ones(n, np.uint8)
            # ---- host-pointer chain
            leg = eh.leg_state(q, qd, R, np.zeros((n, 3)), np.zeros((n, 3)))
            pos, vel, ec = eh.ekf_update(0.0025, mm, ff, R, acc, w, leg["foot_pos_rel"], leg["foot_vel_rel"])
            up = eh.update_plan(mm, st_h["gc"], spd, vel, Rz, R, pos, vd); st_h["gc"] = up["gait_counter"]
            cur, kin = eh.swing_legs(Rz, leg["foot_pos_abs"], st_h["gc"], up["foot_pos_target_rel"], st_h["start"], st_h["rl"], st_h["tl"])
            ctr = eh.contact_terrain(st_h["gc"], up["plan_contacts"], ff, leg["foot_pos_abs"], pos[:, 2],
                    st_h["pitch"]); st_h["pitch"] = ctr["root_euler_d_pitch"]
            tick = scen.pack_tick(eul, pos, w, vel, np.c_[np.zeros(n), st_h["pitch"], eul[:, 2]], vd, wd, np.full(n, 0.3))
            sol = eh.solve_ticks(tick, R, leg["foot_pos_abs"], ctr["contacts"])
            st_h["tau"] = eh.joint_torques(act, ctr["contacts"], leg["Jb"], sol["grf"], kin, km, tg, st_h["tau"])
            # ---- device-pointer chain (same inputs uploaded once per tick, everything else stays on the GPU)
            with torch.cuda.stream(st):
                d = {k: T(v) for k, v in dict(q=q, qd=qd, R=R, Rz=Rz, acc=acc, w=w, ff=ff, mm=mm, vd=vd, wd=wd, spd=spd, tg=tg, act=act,
                        eul=eul, z0=np.zeros((n, 3))).items()}
                o = {k: torch.zeros((n, m), dtype=torch.float64, device=dev) for k,
                        m in dict(rel=12, Jb=36, vrel=12, pabs=12, vabs=12, pw=12, vw=12, pos=3, vel=3, trel=12, tabs=12,
                                                                                             tworld=12, cur=12, kin=12, rec=12,
                                                                                                     grf=12).items()}
                ec_d = torch.zeros((n, 4), dtype=torch.uint8, device=dev); pc_d = torch.zeros((n, 4), dtype=torch.uint8,
                        device=dev); ct_d = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
                ta_d = torch.zeros(n, dtype=torch.float64, device=dev); it_d = torch.zeros(n, dtype=torch.int32,
                        device=dev); stt_d = torch.zeros(n, dtype=torch.int32, device=dev)
                L = ed.lib
                assert L.a1mpc_leg_state_batch_device(ed._h, n, ptr(d["q"]), ptr(d["qd"]), ptr(d["R"]), ptr(d["z0"]), ptr(d["z0"]),
                        dp_(fix), dp_(opt), ptr(o["rel"]), ptr(o["Jb"]),
                                                      ptr(o["vrel"]), ptr(o["pabs"]), ptr(o["vabs"]), ptr(o["pw"]), ptr(o["vw"]), sp) == 0
                assert L.a1mpc_ekf_update_batch_device(ed._h, n, 0.0025, 1, ptr(d["mm"]), ptr(d["ff"]), ptr(d["R"]), ptr(d["acc"]),
                        ptr(d["w"]), ptr(o["rel"]), ptr(o["vrel"]),
                                                       ptr(o["pos"]), ptr(o["vel"]), ptr(ec_d), sp) == 0
                assert L.a1mpc_update_plan_batch_device(ed._h, C.byref(gait), n, ptr(d["mm"]), ptr(st_d["gc"]), ptr(d["spd"]),
                        ptr(o["vel"]), ptr(d["Rz"]), ptr(d["R"]), ptr(o["pos"]),
                                                        ptr(d["vd"]), ptr(pc_d), ptr(o["trel"]), ptr(o["tabs"]), ptr(o["tworld"]), sp) == 0
                assert L.a1mpc_swing_legs_batch_device(ed._h, n, 120.0, 0.0025, ptr(d["Rz"]), ptr(o["pabs"]), ptr(st_d["gc"]),
                        ptr(o["trel"]), dp_(kp), dp_(kd), ptr(st_d["start"]),
                                                       ptr(st_d["rl"]), ptr(st_d["tl"]), ptr(o["cur"]), ptr(o["kin"]), sp) == 0
                pz = o["pos"][:, 2].contiguous()
                assert L.a1mpc_contact_terrain_batch_device(ed._h, C.byref(ccfg), n, ptr(st_d["gc"]), ptr(pc_d), ptr(d["ff"]),
                        ptr(o["pabs"]), ptr(pz), ptr(st_d["pitch"]), ptr(ct_d),
                                                            ptr(o["rec"]), ptr(ta_d), sp) == 0
                zc = torch.zeros(n, dtype=torch.float64, device=dev)
                tick_d = torch.cat([d["eul"], o["pos"], d["w"], o["vel"], torch.stack([zc, st_d["pitch"], d["eul"][:, 2]], 1), d["vd"],
                        d["wd"], torch.full((n, 1), 0.3, dtype=torch.float64, device=dev)], 1).contiguous()
                assert L.a1mpc_solve_batch_ticks_device(ed._h, n, ptr(tick_d), ptr(d["R"]), ptr(o["pabs"]), ptr(ct_d), ptr(o["grf"]),
                        None, ptr(it_d), ptr(stt_d), sp) == 0
                assert L.a1mpc_joint_torques_batch_device(ed._h, n, ptr(d["act"]), ptr(ct_d), ptr(o["Jb"]), ptr(o["grf"]), ptr(o["kin"]),
                        dp_(km), ptr(d["tg"]), ptr(st_d["tau"]), sp) == 0
            st.synchronize()
            assert np.array_equal(st_d["tau"].cpu().numpy(), st_h["tau"]), (t, np.abs(st_d["tau"].cpu().numpy() - st_h["tau"]).max())
            assert np.array_equal(o["pos"].cpu().numpy(), pos) and np.array_equal(ct_d.cpu().numpy(),
                    ctr["contacts"]) and np.array_equal(it_d.cpu().numpy(), sol["iters"])


def tick_inputs(scen, rng, n):
    """random sensors / commands of one control tick (what test_device_pointer_tick_matches_host_pointer_tick feeds the chain)"""
    eul = rng.normal(0, 0.05, (n, 3)); eul[:, 2] = rng.uniform(-1, 1, n)
    return dict(joint_pos=np.tile([0.0, 0.8, -1.6], (n, 4)) + rng.normal(0, 0.1, (n, 12)), joint_vel=rng.normal(0, 1, (n, 12)),
                R_world=scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9), R_z=scen.rot_zyx(0 * eul[:, 0], 0 * eul[:, 0], eul[:, 2]).reshape(n, 9),
                root_euler=eul, root_ang_vel=rng.normal(0, 0.2, (n, 3)), imu_acc=np.array([0, 0, 9.81]) + rng.normal(0, 0.2, (n, 3)),
                imu_ang_vel=rng.normal(0, 0.2, (n, 3)), foot_force=rng.uniform(0, 120, (n, 4)), movement_mode=np.ones(n, np.uint8),
                mpc_active=(rng.random(n) < 0.9).astype(np.uint8), root_lin_vel_d=np.c_[rng.uniform(-0.4, 0.4, (n, 2)), np.zeros(n)],
                root_ang_vel_d=np.c_[np.zeros((n, 2)), rng.uniform(-0.4, 0.4, n)], root_pos_d_z=np.full(n, 0.3), gait_counter_speed=np.full((n, 4), 2.0),
                torques_gravity=rng.normal(0, 0.5, (n, 12)))


TICK_STATE = dict(gait_counter=4, foot_pos_start=12, foot_pos_rel_last_time=12, foot_pos_target_last_time=12, root_euler_d=3, joint_torques=12, root_pos=3,
                  root_lin_vel=3)
TICK_OUT_F64 = dict(foot_pos_rel=12, j_foot_blocks=36, foot_vel_rel=12, foot_pos_abs=12, foot_vel_abs=12, foot_pos_world=12, foot_vel_world=12, foot_pos_target_rel=12,
                    foot_pos_target_abs=12, foot_pos_target_world=12, foot_pos_cur=12, foot_forces_kin=12, foot_pos_recent_contact=12, terrain_angle=1, grf=12)


@pytest.mark.parametrize("n,warm", [(300, 1), (64, 2), (4096, 1)])
def test_control_tick_one_call_matches_the_seven_entry_chain(pkg, scen, n, warm):
    """a1mpc_control_tick_device -- leg state, EKF, gait plan, swing legs, contacts / terrain, MPC from tick records and the joint torques in ONE C call,
    N3 inside the MPC kernel's output stage -- against the seven *_device entry points chained by hand on a second handle: every output and every carried state bit for
    bit, four ticks.  n = 300: the fused kernel from the first tick; 64: the latency kernel, update path; 4096: the split pipeline on the first tick (torques by their
    own launch), then the fused kernel in the order of the previous tick's costs (torques in the output stage)."""
    import torch
    rng = np.random.default_rng(2025 + n)
    P = scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS
    cfg = pkg.make_config(P, 10, warm_start=warm)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    dp_ = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    E = pkg.engine
    with pkg.Engine(cfg, n, 0) as e1, pkg.Engine(cfg, n, 0) as e7:
        prm = E.TickParams(); e1.lib.a1mpc_default_tick_params(C.byref(prm))
        assert np.allclose(np.array(prm.rho_fix).reshape(4, 5), e1.A1_RHO_FIX) and list(prm.km_foot) == [0.1, 0.1, 0.04]
        kp = np.array(prm.kp_foot); kd = np.array(prm.kd_foot); km = np.array(prm.km_foot); fix = np.array(prm.rho_fix); opt = np.array(prm.rho_opt)
        st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
        init = dict(gait_counter=np.tile([0.0, 120.0, 120.0, 0.0], (n, 1)), root_pos=np.tile([0.0, 0.0, 0.3], (n, 1)))
        state = [{k: T(init.get(k, np.zeros((n, m)))) for k, m in TICK_STATE.items()} for _ in range(2)]
        outs = [{k: torch.zeros((n, m) if m > 1 else (n,), dtype=torch.float64, device=dev) for k, m in TICK_OUT_F64.items()} for _ in range(2)]
        u8 = [{k: torch.zeros((n, 4), dtype=torch.uint8, device=dev) for k in ("estimated_contacts", "plan_contacts", "contacts")} for _ in range(2)]
        i32 = [{k: torch.zeros(n, dtype=torch.int32, device=dev) for k in ("iters", "status")} for _ in range(2)]
        fused_seen = []
        for t in range(4):
            inp = {k: T(v) for k, v in tick_inputs(scen, rng, n).items()}
            # ---- one call
            bf = E.TickBuffers()
            for k in E.TICK_BUFFER_FIELDS:
                src = inp if k in inp else state[0] if k in state[0] else outs[0] if k in outs[0] else u8[0] if k in u8[0] else i32[0]
                setattr(bf, k, src[k].data_ptr())
            e1.control_tick_device(prm, bf, n, stream=st.cuda_stream)
            fused_seen.append(e1.last_control_tick_ms()[1])
            # ---- the chain
            s7, o7, b7, j7, L, H_ = state[1], outs[1], u8[1], i32[1], e7.lib, e7._h
            rcs = [L.a1mpc_leg_state_batch_device(H_, n, ptr(inp["joint_pos"]), ptr(inp["joint_vel"]), ptr(inp["R_world"]), ptr(s7["root_pos"]), ptr(s7["root_lin_vel"]), dp_(fix),
                                                  dp_(opt), ptr(o7["foot_pos_rel"]), ptr(o7["j_foot_blocks"]), ptr(o7["foot_vel_rel"]), ptr(o7["foot_pos_abs"]),
                                                  ptr(o7["foot_vel_abs"]), ptr(o7["foot_pos_world"]), ptr(o7["foot_vel_world"]), sp),
                   L.a1mpc_ekf_update_batch_device(H_, n, prm.control_dt, 1, ptr(inp["movement_mode"]), ptr(inp["foot_force"]), ptr(inp["R_world"]), ptr(inp["imu_acc"]),
                                                   ptr(inp["imu_ang_vel"]), ptr(o7["foot_pos_rel"]), ptr(o7["foot_vel_rel"]), ptr(s7["root_pos"]), ptr(s7["root_lin_vel"]),
                                                   ptr(b7["estimated_contacts"]), sp),
                   L.a1mpc_update_plan_batch_device(H_, C.byref(prm.gait), n, ptr(inp["movement_mode"]), ptr(s7["gait_counter"]), ptr(inp["gait_counter_speed"]),
                                                    ptr(s7["root_lin_vel"]), ptr(inp["R_z"]), ptr(inp["R_world"]), ptr(s7["root_pos"]), ptr(inp["root_lin_vel_d"]),
                                                    ptr(b7["plan_contacts"]), ptr(o7["foot_pos_target_rel"]), ptr(o7["foot_pos_target_abs"]), ptr(o7["foot_pos_target_world"]), sp),
                   L.a1mpc_swing_legs_batch_device(H_, n, prm.gait.counter_per_swing, prm.control_dt, ptr(inp["R_z"]), ptr(o7["foot_pos_abs"]), ptr(s7["gait_counter"]),
                                                   ptr(o7["foot_pos_target_rel"]), dp_(kp), dp_(kd), ptr(s7["foot_pos_start"]), ptr(s7["foot_pos_rel_last_time"]),
                                                   ptr(s7["foot_pos_target_last_time"]), ptr(o7["foot_pos_cur"]), ptr(o7["foot_forces_kin"]), sp)]
            with torch.cuda.stream(st):
                pz = s7["root_pos"][:, 2].contiguous(); pitch = s7["root_euler_d"][:, 1].contiguous()
            rcs.append(L.a1mpc_contact_terrain_batch_device(H_, C.byref(prm.contact), n, ptr(s7["gait_counter"]), ptr(b7["plan_contacts"]), ptr(inp["foot_force"]),
                                                            ptr(o7["foot_pos_abs"]), ptr(pz), ptr(pitch), ptr(b7["contacts"]), ptr(o7["foot_pos_recent_contact"]),
                                                            ptr(o7["terrain_angle"]), sp))
            with torch.cuda.stream(st):
                s7["root_euler_d"][:, 1] = pitch
                tick = torch.cat([inp["root_euler"], s7["root_pos"], inp["root_ang_vel"], s7["root_lin_vel"], s7["root_euler_d"], inp["root_lin_vel_d"], inp["root_ang_vel_d"],
                                  inp["root_pos_d_z"].reshape(n, 1)], 1).contiguous()
            rcs.append(L.a1mpc_solve_batch_ticks_device(H_, n, ptr(tick), ptr(inp["R_world"]), ptr(o7["foot_pos_abs"]), ptr(b7["contacts"]), ptr(o7["grf"]), None,
                                                        ptr(j7["iters"]), ptr(j7["status"]), sp))
            rcs.append(L.a1mpc_joint_torques_batch_device(H_, n, ptr(inp["mpc_active"]), ptr(b7["contacts"]), ptr(o7["j_foot_blocks"]), ptr(o7["grf"]), ptr(o7["foot_forces_kin"]),
                                                          dp_(km), ptr(inp["torques_gravity"]), ptr(s7["joint_torques"]), sp))
            assert not any(rcs), rcs
            st.synchronize()
            for grp in (state, outs, u8, i32):
                for k in grp[0]:
                    a, b = grp[0][k].cpu().numpy(), grp[1][k].cpu().numpy()
                    assert np.array_equal(a, b, equal_nan=True), (t, k, np.abs(a.astype(float) - b.astype(float)).max())
            assert (i32[0]["status"].cpu().numpy() == 1).all() and np.abs(state[0]["joint_torques"].cpu().numpy()).max() > 0.1
        assert fused_seen == ([True] * 4 if n <= 2048 else [False, True, True, True]), fused_seen


# ------------------------------------------------------------------------------------------------------------ round 2
def test_bench_batch_all_4096_qps_vs_oracle(pkg, oracle, scen):
    """VERDICT r1 task 2: EVERY QP of the bench configuration (BASELINE configs[2]: 4096 x h10, the bench seed) against the oracle."""
    sc = scen.config3_random_flat(nb=4096)
    with _engine(pkg, sc, 4096, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    r = compare(out, oracle_batch(oracle, sc), min_same=1.0)
    print("4096 x h10:", r)


@pytest.mark.parametrize("gen,n", [("config4_random_h16", 8192), ("config5_divergent", 8192)])
def test_full_8192_h16_h20_vs_oracle(pkg, oracle, scen, gen, n):
    """VERDICT r1 task 2: one full 8192 x h16 (configs[3] per-GPU share) and one 8192 x h20 (configs[4] shape) comparison, every QP."""
    sc = getattr(scen, gen)(nb=n)
    with _engine(pkg, sc, n, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    r = compare(out, oracle_batch(oracle, sc), min_same=1.0)
    print(gen, r)


def _strided_inputs(scen, rng, h, nb, feet, cont):
    sc = scen.config3_random_flat(nb=nb, horizon=h)
    p = sc["params"]; foot = sc["foot"]; contact = sc["contact"]; fs = cs = 0
    if feet:
        vd = rng.uniform(-0.6, 0.6, (nb, 1, 1, 3))
        foot = (sc["foot"].reshape(nb, 1, 4, 3) - vd * p["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(nb, h * 12); fs = 12
    if cont:
        sw = rng.integers(0, h + 1, (nb, 4)); first = rng.integers(0, 2, (nb, 4))
        contact = np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :],
                1 - first[:, None, :]).astype(np.uint8).reshape(nb, h * 4); cs = 4
    return sc, np.ascontiguousarray(foot), fs, np.ascontiguousarray(contact), cs


@pytest.mark.parametrize("h,nb,feet,cont", [(10, 300, True, True), (10, 128, True, False), (10, 128, False, True), (16, 96, True, True),
        (20, 64, True, True)])
def test_per_step_feet_and_contact_schedules(pkg, oracle, scen, h, nb, feet, cont):
    """b' (VERDICT r1): a1mpc_solve_batch_strided -- per-step B_d (S/ConvexMpc.h:74 B_mat_d_list, S/test/test_mpc.cpp:106-122) and per-step
    contact schedules -- vs the oracle's strided formation (which oracle/_ref pins to the reference's ConvexMpc for per-step feet)."""
    rng = np.random.default_rng(1000 + h + 2 * feet + cont)
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, feet, cont)
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        out = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True)
        bc = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], sc["foot"], 0, sc["contact"], 0, want_u=True)
        fast = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    assert np.array_equal(bc["u"], fast["u"]) and np.array_equal(bc["iters"], fast["iters"])   # (0, 0) is the fast path, bit for bit
    pr = oracle.mpc_params(h, **{k: sc["params"][k] for k in ("dt", "mu", "fz_min", "fz_max", "q", "r", "mass",
            "inertia")}); st = oracle.default_settings()
    worst = 0.0
    for b in range(0, nb, 3):
        r = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=fs, contact_stride=cs)
        assert out["iters"][b] == r["info"].iters and out["status"][b] == r["info"].status, (b, out["iters"][b], r["info"].iters)
        worst = max(worst, np.abs(out["u"][b] - r["u"]).max(), np.abs(out["grf"][b] - r["grf"]).max())
    assert worst <= TOL_FORCE_N, worst
    if cont:   # a leg that is in swing at step t carries no force at step t
        u = out["u"].reshape(nb, h, 4, 3); c = contact.reshape(nb, h, 4)
        assert np.abs(u[c == 0]).max() < 1.0


@pytest.mark.parametrize("h,nb", [(10, 1), (10, 48), (16, 6), (20, 5), (10, 4000), (16, 2100), (20, 1700)])
def test_update_path_on_the_general_path(pkg, oracle, scen, h, nb):
    """Round 5 (VERDICT r4 missing 3 / item 5): warm_start = 2 -- the reference's per-tick OSQP update path -- on a STRIDED tick sequence: per-step feet that drift by
    -v_d dt per horizon step (S/test/test_mpc.cpp:112-115) and the gait's contact schedule over the horizon.  a1mpc_last_warm_start_mode reports 2, and every tick has the
    oracle's iteration count, status and forces (orc_mpc_solve_update_strided: the same persistent-solver semantics on the QP those inputs form) -- through a contact switch.
    Round 6: batches beyond the resident rows of the fused general kernel (4000 / 2100 / 1700 QPs at h = 10 / 16 / 20) follow the update path too -- a sample of the robots
    is chained through the oracle, tick for tick."""
    seq = scen.config2_trot_sequence(70, horizon=h)
    pr = oracle_params(oracle, seq); st = oracle.default_settings(warm_start=1); dt = seq["params"]["dt"]
    rng = np.random.default_rng(77 + h)
    checked = list(range(nb)) if nb <= 48 else sorted(set(list(range(0, nb, max(1, nb // 24))) + [nb - 1]))   # the robots whose ticks the oracle follows (every robot of the small batches)
    carries = {b: oracle.update_carry(h) for b in checked}
    ticks = (list(range(0, 5)) + list(range(56, 63))) if nb <= 48 else (list(range(0, 3)) + list(range(58, 62)))
    worst = 0.0
    with _engine(pkg, seq, nb, warm_start=2) as eng:
        for i, k in enumerate(ticks):
            x0 = np.repeat(seq["x0"][k:k + 1], nb, 0); x0[:, :12] += rng.normal(0, 1e-3, (nb, 12)) * (np.arange(nb)[:, None] > 0)
            vd = np.c_[np.full(nb, 0.3), 0.05 * np.sin(k + np.arange(nb)), np.zeros(nb)]
            foot = (seq["foot"][k].reshape(1, 1, 4, 3) - vd.reshape(nb, 1, 1, 3) * dt * np.arange(h).reshape(1, h, 1, 1)).reshape(nb, 12 * h)
            phase = (k + np.arange(h)) // 60 % 2 == 0
            contact = np.repeat(np.where(phase[:, None], [1, 0, 0, 1], [0, 1, 1, 0]).astype(np.uint8).reshape(1, 4 * h), nb, 0)
            xref = np.repeat(seq["xref"][k:k + 1], nb, 0); R = np.repeat(seq["R"][k:k + 1], nb, 0)
            out = eng.solve_strided(x0, xref, R, foot, 12, contact, 4)
            assert eng.last_warm_start_mode() == 2
            for b in checked:
                o = oracle.mpc_solve_update(pr, st, x0[b], xref[b], R[b], foot[b], contact[b], carries[b], foot_stride=12, contact_stride=4)
                assert out["iters"][b] == o["info"].iters and out["status"][b] == o["info"].status, (h, k, b, out["iters"][b], o["info"].iters)
                worst = max(worst, np.abs(out["grf"][b] - o["grf"]).max())
    assert worst <= 1e-7, worst
    print(f"h{h} x {nb}: {len(ticks)} strided update-path ticks, worst |dGRF| {worst:.1e} N")


@pytest.mark.parametrize("h,nb", [(10, 1), (10, 40), (16, 5)])
def test_update_path_across_a_switch_between_the_fast_and_the_general_path(pkg, oracle, scen, h, nb):
    """a handle on warm_start = 2 whose caller alternates between step-invariant feet (fast path) and per-step feet (general path) from tick to tick.  The
    two paths encode the carried pattern signature differently; a switch keeps the UPDATE path (the reference's persistent solver takes osqp_update_P as long as the
    dense Hessian keeps its pattern, which it does on either path for inputs in general position).  Every tick vs the oracle's persistent-solver semantics on the QP the
    tick's inputs form (orc_mpc_solve_update, strided or not): same iteration count, status, forces."""
    seq = scen.config2_trot_sequence(30, horizon=h)
    pr = oracle_params(oracle, seq); st = oracle.default_settings(warm_start=1); dt = seq["params"]["dt"]
    rng = np.random.default_rng(900 + h)
    carries = [oracle.update_carry(h) for _ in range(nb)]
    worst = 0.0
    pattern = [0, 1, 1, 0, 1, 0, 0, 1]   # 1 = per-step feet this tick
    with _engine(pkg, seq, nb, warm_start=2) as eng:
        for k, gen in enumerate(pattern):
            x0 = np.repeat(seq["x0"][k:k + 1], nb, 0); x0[:, :12] += rng.normal(0, 1e-3, (nb, 12)) * (np.arange(nb)[:, None] > 0)
            xref = np.repeat(seq["xref"][k:k + 1], nb, 0); R = np.repeat(seq["R"][k:k + 1], nb, 0)
            contact = np.repeat(seq["contact"][k:k + 1], nb, 0)
            if gen:
                vd = np.c_[np.full(nb, 0.3), 0.05 * np.sin(k + np.arange(nb)), np.zeros(nb)]
                foot = (seq["foot"][k].reshape(1, 1, 4, 3) - vd.reshape(nb, 1, 1, 3) * dt * np.arange(h).reshape(1, h, 1, 1)).reshape(nb, 12 * h)
                out = eng.solve_strided(x0, xref, R, foot, 12, contact, 0)
            else:
                foot = np.repeat(seq["foot"][k:k + 1], nb, 0)
                out = eng.solve(x0, xref, R, foot, contact)
            assert eng.last_warm_start_mode() == 2
            for b in range(nb):
                o = oracle.mpc_solve_update(pr, st, x0[b], xref[b], R[b], foot[b], contact[b], carries[b], foot_stride=12 if gen else 0, contact_stride=0)
                assert out["iters"][b] == o["info"].iters and out["status"][b] == o["info"].status, (h, k, gen, b, out["iters"][b], o["info"].iters)
                worst = max(worst, np.abs(out["grf"][b] - o["grf"]).max())
    assert worst <= 1e-7, worst


def test_general_path_latency_kernel(pkg, oracle, scen):
    """Round 5: a handful of general-path QPs at h = 10 (<= 256: the reference's own use of the interface is ONE, S/test/test_mpc.cpp:106-122) run one per wavefront with
    the four rows sharing the set-up (a1mpc_solve_gen_coop_kernel).  Bit for bit what the same QPs give inside a batch of 300 (the fused general kernel: main / twin
    pairs), the oracle's strided formation on every QP, and the same on the update path (warm_start = 2) over four ticks."""
    h, nb, small = 10, 300, 9
    rng = np.random.default_rng(4242)
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, True, True)
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        big = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True)
        lat = eng.solve_strided(sc["x0"][:small], sc["xref"][:small], sc["R"][:small], foot[:small], fs, contact[:small], cs, want_u=True)
        one = eng.solve_strided(sc["x0"][:1], sc["xref"][:1], sc["R"][:1], foot[:1], fs, contact[:1], cs, want_u=True)
    for k in ("u", "grf", "iters", "status"):
        assert np.array_equal(big[k][:small], lat[k]) and np.array_equal(big[k][:1], one[k]), k
    p = sc["params"]
    pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"]); st = oracle.default_settings()
    for b in range(small):
        o = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=fs, contact_stride=cs)
        assert o["info"].iters == lat["iters"][b] and o["info"].status == lat["status"][b] and np.abs(o["grf"] - lat["grf"][b]).max() < 1e-5, b
    # update path: a batch of 300 (fused general kernel) and its first 9 robots alone (latency kernel) tick side by side
    with _engine(pkg, sc, nb, warm_start=2) as e_big, _engine(pkg, sc, small, warm_start=2) as e_lat:
        x0 = sc["x0"].copy()
        for t in range(4):
            x0[:, :12] += np.random.default_rng(t).normal(0, 2e-3, (nb, 12))
            a = e_big.solve_strided(x0, sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True)
            b = e_lat.solve_strided(x0[:small], sc["xref"][:small], sc["R"][:small], foot[:small], fs, contact[:small], cs, want_u=True)
            assert np.array_equal(a["u"][:small], b["u"]) and np.array_equal(a["iters"][:small], b["iters"]), t
        assert e_lat.last_warm_start_mode() == 2 and e_big.last_warm_start_mode() == 2


@pytest.mark.parametrize("h,nb", [(10, 4000), (16, 2100), (20, 1700)])
def test_general_path_split_pipeline(pkg, oracle, scen, h, nb):
    """a general-path batch beyond its resident rows runs the general path's own set-up kernel + persistent main / twin pairs on a queue
    (the hand-off
    record carries B~w_t of every step): bit for bit the fused general-path kernel (the first 200 QPs solved alone), the oracle's strided
    formation on a
    sample, and a re-solve in history order"""
    rng = np.random.default_rng(5000 + h)
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, True, True)
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        out = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True)
        # queue now ordered by the first solve's costs
        again = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True)
        small = eng.solve_strided(sc["x0"][:200], sc["xref"][:200], sc["R"][:200], foot[:200], fs, contact[:200], cs, want_u=True)
    assert np.array_equal(out["u"], again["u"]) and np.array_equal(out["iters"], again["iters"])
    assert np.array_equal(out["u"][:200], small["u"]) and np.array_equal(out["iters"][:200],
            small["iters"]) and np.array_equal(out["grf"][:200], small["grf"])
    pr = oracle.mpc_params(h, **{k: sc["params"][k] for k in ("dt", "mu", "fz_min", "fz_max", "q", "r", "mass",
            "inertia")}); st = oracle.default_settings()
    worst = 0.0
    for b in range(0, nb, nb // 50):
        r = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=fs, contact_stride=cs)
        assert out["iters"][b] == r["info"].iters and out["status"][b] == r["info"].status, (b, out["iters"][b], r["info"].iters)
        worst = max(worst, np.abs(out["u"][b] - r["u"]).max(), np.abs(out["grf"][b] - r["grf"]).max())
    assert worst <= TOL_FORCE_N, worst


@pytest.mark.parametrize("h,nb", [(10, 4500), (10, 700), (10, 1), (16, 1200), (16, 1500), (20, 2300)])
def test_contact_schedule_alone_stays_on_the_fast_kernels(pkg, oracle, scen, h, nb):
    """a per-step contact schedule with step-invariant feet (contact_stride = 4, foot_stride = 0, no yaw_A): the fast kernels take it
    (set-up
    kernel + persistent twin rows, fused kernel, latency kernel by batch size) -- vs the oracle's strided formation on a sample, and vs the
    general path on every QP (a yaw_A equal to the state's yaw forces the general path onto the same QP)."""
    rng = np.random.default_rng(4000 + h + nb)
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, False, True)
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        out = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], sc["foot"], 0, contact, 4, want_u=True)
        gen = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], sc["foot"], 0, contact, 4, want_u=True, yaw_A=sc["x0"][:, 2].copy())
        ms_fast = None
    assert (out["status"] == 1).all() and np.array_equal(out["iters"], gen["iters"]) and np.abs(out["u"] - gen["u"]).max() <= 1e-7
    pr = oracle.mpc_params(h, **{k: sc["params"][k] for k in ("dt", "mu", "fz_min", "fz_max", "q", "r", "mass",
            "inertia")}); st = oracle.default_settings()
    worst = 0.0
    for b in range(0, nb, max(1, nb // 40)):
        r = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], contact[b], foot_stride=0, contact_stride=4)
        assert out["iters"][b] == r["info"].iters and out["status"][b] == r["info"].status, (b, out["iters"][b], r["info"].iters)
        worst = max(worst, np.abs(out["u"][b] - r["u"]).max(), np.abs(out["grf"][b] - r["grf"]).max())
    assert worst <= TOL_FORCE_N, worst
    u = out["u"].reshape(nb, h, 4, 3); c = contact.reshape(nb, h, 4)
    assert np.abs(u[c == 0]).max() < 1.0   # a leg in swing at step t carries no force at step t


def test_failed_tick_leaves_a_cold_start_behind(pkg, oracle, scen):
    """warm start ON, a NaN tick for some robots -> status -7 and zero GRFs for them at that tick, and at the NEXT tick
    they are solved again from cold iterates (x = y = 0) instead of staying NaN for ever -- with the rho the solver had reached, as OSQP's
    store_solution() -> cold_start() leaves it (the reference ignores the return code, its next tick runs with that rho)."""
    n = 64
    sc = scen.config3_random_flat(nb=n)
    bad = np.zeros(n, bool); bad[[3, 17, 40]] = True
    with _engine(pkg, sc, n, warm_start=1) as eng:
        eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        _, _, rho0 = eng.get_warm_start(n)
        x0 = sc["x0"].copy(); x0[bad, 4] = np.nan
        o1 = eng.solve(x0, sc["xref"], sc["R"], sc["foot"], sc["contact"])
        assert (o1["status"][bad] == -7).all() and (o1["grf"][bad] == 0).all() and (o1["status"][~bad] == 1).all()
        wx, wy, rho = eng.get_warm_start(n)
        assert (wx[bad] == 0).all() and (wy[bad] == 0).all() and np.isfinite(wx).all() and np.isfinite(wy).all()
        # the failed solve never got to adapt: the rho it was started with stays
        assert np.array_equal(rho[bad], rho0[bad]) and (rho0[bad] > 0).all()
        o2 = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    assert (o2["status"] == 1).all()
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    for i in np.flatnonzero(bad):   # cold iterates + the carried rho: the oracle started the same way
        r = oracle.mpc_solve(pr, st, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i], warm_x=np.zeros(120),
                warm_y=np.zeros(200), warm_rho=rho[i])
        assert o2["iters"][i] == r["info"].iters and np.abs(o2["u"][i] - r["u"]).max() <= TOL_FORCE_N, i


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


def test_calls_on_different_streams_are_ordered(pkg, scen):
    """two device-pointer solves of one handle issued on two different streams must not overlap on the handle's
    scratch (prepared-state records, queue counter): results equal the same two solves issued on one stream."""
    import torch
    n = 4096
    sc = scen.config3_random_flat(nb=n); sc2 = scen.config3_random_flat(nb=n, seed=77)
    dev = torch.device("cuda:0")
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    ins = [[t(s["x0"]), t(s["xref"]), t(s["R"]), t(s["foot"]), t(s["contact"], torch.uint8)] for s in (sc, sc2)]
    with _engine(pkg, sc, n, warm_start=0) as eng:
        ref = []
        for k in range(2):
            g = torch.zeros(n, 12, dtype=torch.float64, device=dev)
            eng.solve_device(n, *ins[k], g); torch.cuda.synchronize(); ref.append(g.cpu().numpy())
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for rep in range(3):
            g1 = torch.zeros(n, 12, dtype=torch.float64, device=dev); g2 = torch.zeros(n, 12, dtype=torch.float64, device=dev)
            eng.solve_device(n, *ins[0], g1, stream=s1.cuda_stream)
            eng.solve_device(n, *ins[1], g2, stream=s2.cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(g1.cpu().numpy(), ref[0]) and np.array_equal(g2.cpu().numpy(), ref[1]), rep


@pytest.mark.parametrize("h", [10, 16, 20])
def test_gpu_formed_dense_qp_equals_reference_ConvexMpc(pkg, oracle, scen, h):
    """a1mpc_form_qp_batch (the GPU's implicit Hessian written out entry by entry) vs S/ConvexMpc.cpp compiled verbatim (oracle/_ref), and
    vs the
    oracle: P, g, l, u for broadcast and per-step feet / contacts.  This compares the engine's formation with the REFERENCE directly,
    not through iterates."""
    import ref as REF
    if not REF.build():
        pytest.skip("oracle/_ref not available")
    rng = np.random.default_rng(h)
    nb = 6
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, True, True)
    p = sc["params"]
    pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        bc = eng.form_qp(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        ps = eng.form_qp(sc["x0"], sc["xref"], sc["R"], foot, sc["contact"], foot_stride=12)
        pc = eng.form_qp(sc["x0"], sc["xref"], sc["R"], foot, contact, foot_stride=12, contact_stride=4)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    for b in range(nb):
        for out, f, fstr in ((bc, sc["foot"][b], 0), (ps, foot[b], 12)):
            r = REF.convex_mpc_form(h, p["q"], p["r"], sc["x0"][b][:3], p["mass"], p["inertia"], sc["R"][b], f, sc["contact"][b],
                    sc["x0"][b], sc["xref"][b], p["dt"],
                                    foot_stride=fstr)
            assert rel(out["P"][b], r["P"]) <= 1e-12 and rel(out["g"][b], r["g"]) <= 1e-10, (b, fstr, rel(out["P"][b], r["P"]),
                    rel(out["g"][b], r["g"]))
            assert np.array_equal(out["l"][b], r["l"]) and np.array_equal(out["u"][b], r["u"])
        P, g, A, l, u, _ = oracle.mpc_form(pr, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=12,
                contact_stride=4)
        assert rel(pc["P"][b], P) <= 1e-12 and rel(pc["g"][b], g) <= 1e-10 and np.array_equal(pc["l"][b],
                l) and np.array_equal(pc["u"][b], u)


def test_terrain_block_alone(pkg, oracle, scen):
    """a1mpc_terrain_batch (the terrain block of compute_grf with the caller's foot_pos_recent_contact)
    vs the full N2b entry fed the same way."""
    rng = np.random.default_rng(3)
    n = 200
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    base = np.outer([0.2, 0.2, -0.2, -0.2], [1.0, 0.0, 0.3]).reshape(12) + np.outer([1, -1, 1, -1], [0.0, 0.13, 0.0]).reshape(12)
    pitch_a = np.zeros(n); pitch_b = np.zeros(n)
    # all feet in contact: every recent-contact filter updates
    gcs = np.tile([0.0, 0.0, 0.0, 0.0], (n, 1)); plan = np.ones((n, 4), np.uint8)
    with pkg.Engine(cfg, n, 0) as full, pkg.Engine(cfg, n, 0) as only:
        for t in range(130):
            foot = base + rng.normal(0, 0.03, (n, 12)) + np.tile([0.0, 0.0, -0.3], 4); z = np.where(rng.random(n) < 0.9, 0.3,
                    0.05); ff = rng.uniform(0, 80, (n, 4))
            o = full.contact_terrain(gcs, plan, ff, foot, z, pitch_a); pitch_a = o["root_euler_d_pitch"]
            pitch_b, ta = only.terrain(o["foot_pos_recent_contact"], z, pitch_b)
            assert np.array_equal(ta, o["terrain_angle"]) and np.array_equal(pitch_b, pitch_a), t
    assert np.abs(pitch_a).max() > 0.05


def test_native_sharded_handle_two_shards_on_one_gpu(pkg, scen):
    """a1mpc_sharded_* (SURVEY 8b device = -1, 8e): the batch cut into contiguous shards behind one handle.  The test box has one GPU, so
    the
    pinned-copy transport runs two (three) shards on device 0 -- results must equal the single-handle solve bit for bit, ragged sizes
    included;
    the RCCL transport is created on the one device (communicator set-up, root staging; no peer to talk to)."""
    sc = scen.config3_random_flat(nb=4097)
    cfg = pkg.make_config(sc["params"], 10, warm_start=0)
    with pkg.Engine(cfg, 4097, 0) as eng:
        ref = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    for devs in ([0, 0], [0, 0, 0]):
        with pkg.ShardedEngine(cfg, 4097, devices=devs, transport=0) as sh:
            assert sh.info()["n_shards"] == len(devs)
            for n in (4097, 5, 1):
                out = sh.solve(sc["x0"][:n], sc["xref"][:n], sc["R"][:n], sc["foot"][:n], sc["contact"][:n])
                assert np.array_equal(out["grf"], ref["grf"][:n]) and np.array_equal(out["iters"],
                        ref["iters"][:n]) and np.array_equal(out["status"], ref["status"][:n]), (devs, n)
    with pkg.ShardedEngine(cfg, 512, devices=None, transport=0) as sh:   # "all visible devices"
        out = sh.solve(sc["x0"][:512], sc["xref"][:512], sc["R"][:512], sc["foot"][:512], sc["contact"][:512])
        assert np.array_equal(out["grf"], ref["grf"][:512])
    with pkg.ShardedEngine(cfg, 512, devices=[0], transport=1) as sh:    # RCCL transport, one rank
        out = sh.solve(sc["x0"][:512], sc["xref"][:512], sc["R"][:512], sc["foot"][:512], sc["contact"][:512])
        assert np.array_equal(out["grf"], ref["grf"][:512]) and sh.info()["transport"] == 1
    with pytest.raises(pkg.A1MpcError):
        pkg.ShardedEngine(cfg, 512, devices=[0, 0], transport=1)       # RCCL needs distinct devices


def test_stage_split_instrumentation(pkg, scen):
    """SURVEY 5 "tracing": form | solve split of the last launch (the reference's t1..t6 stopwatches, S/A1RobotControl.cpp:491-553)"""
    sc = scen.config3_random_flat(nb=8192)
    with _engine(pkg, sc, 8192, warm_start=0) as eng:
        eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        form, solve = eng.last_stage_ms(); total = eng.last_kernel_ms()
        assert form > 0.05 and solve > form and abs(form + solve - total) < 0.05 * total, (form, solve, total)
        eng.solve(sc["x0"][:64], sc["xref"][:64], sc["R"][:64], sc["foot"][:64], sc["contact"][:64])   # fused kernel: not separable
        form, solve = eng.last_stage_ms()
        assert form == 0.0 and solve > 0.0


def test_general_path_warm_started_sequence_and_device_pointers(pkg, oracle, scen):
    """The general path carries the OSQP workspace like the fast path (warm-started ticks with per-step feet / contacts vs the oracle
    chained the
    same way), and its device-pointer entry gives the host entry's numbers."""
    import ctypes as C
    import torch
    h, nb, ticks = 10, 48, 5
    rng = np.random.default_rng(77)
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, True, True)
    p = sc["params"]
    pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"],
            p["inertia"]); st = oracle.default_settings(warm_start=1)
    wx = np.zeros((nb, 12 * h)); wy = np.zeros((nb, 20 * h)); rho = np.zeros(nb)
    with _engine(pkg, sc, nb, warm_start=1) as eng:
        for t in range(ticks):
            x0 = sc["x0"].copy(); x0[:, :12] += rng.normal(0, 0.002, (nb, 12)) * t
            out = eng.solve_strided(x0, sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True)
            for b in range(0, nb, 5):
                r = oracle.mpc_solve(pr, st, x0[b], sc["xref"][b], sc["R"][b], foot[b], contact[b], warm_x=wx[b], warm_y=wy[b],
                        warm_rho=rho[b], foot_stride=fs, contact_stride=cs)
                wx[b], wy[b], rho[b] = r["warm_x"], r["warm_y"], r["rho"]
                assert out["iters"][b] == r["info"].iters, (t, b, out["iters"][b], r["info"].iters)
                assert np.abs(out["u"][b] - r["u"]).max() <= TOL_FORCE_N
    dev = torch.device("cuda:0")
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    d = [T(sc["x0"]), T(sc["xref"]), T(sc["R"]), T(foot), T(contact, torch.uint8)]
    g = torch.zeros(nb, 12, dtype=torch.float64, device=dev); it = torch.zeros(nb, dtype=torch.int32, device=dev); stt = torch.zeros(nb,
            dtype=torch.int32, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        host = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs)
        rc = eng.lib.a1mpc_solve_batch_strided_device(eng._h, nb, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), fs, ptr(d[4]), cs, None,
                ptr(g), None, ptr(it), ptr(stt), None)
        assert rc == 0
        torch.cuda.synchronize()
    assert np.array_equal(g.cpu().numpy(), host["grf"]) and np.array_equal(it.cpu().numpy(), host["iters"])


def test_batch_pipeline_overlaps_batches_and_changes_no_bit(pkg, oracle, scen):
    """a1mpc_pipeline_*: consecutive batches in flight on `depth` handles / HIP streams.  Every batch comes back bit-identical to a lone
    handle's
    solve (and the first one is oracle-checked), slots go round-robin, a fixed slot keeps its warm start, wait / join deliver the outputs,
    and at
    4096 x h10 two batches in flight are faster per batch than one (the next batch runs in the tail of the one before)."""
    import time
    import torch
    n, NB = 4096, 4
    dev = torch.device("cuda:0")
    scs = [scen.config3_random_flat(nb=n, seed=500 + k) for k in range(NB)]
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    ins = [[t(s["x0"]), t(s["xref"]), t(s["R"]), t(s["foot"]), t(s["contact"], torch.uint8)] for s in scs]
    cfg = pkg.make_config(scs[0]["params"], 10, warm_start=0)
    ref = []
    with pkg.Engine(cfg, n, 0) as eng:
        for k in range(NB):
            g = torch.zeros(n, 12, dtype=torch.float64, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
            eng.set_schedule(True)
            eng.solve_device(n, *ins[k], g, None, it); torch.cuda.synchronize()
            ref.append((g.cpu().numpy(), it.cpu().numpy()))
    o = oracle_batch(oracle, scs[0])
    assert np.abs(ref[0][0] - o["grf"]).max() <= TOL_FORCE_N and (ref[0][1] == o["iters"]).all()

    def run(depth, steps):
        outs = [(torch.zeros(n, 12, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
        with pkg.Pipeline(cfg, n, 0, depth=depth) as pipe:
            assert pipe.depth == depth
            slots = [pipe.submit_device(n, *ins[k % NB], outs[k % NB][0], None, outs[k % NB][1]) for k in range(NB)]
            assert slots == [k % depth for k in range(NB)]
            pipe.wait()
            for k in range(NB):
                assert np.array_equal(outs[k][0].cpu().numpy(), ref[k][0]) and np.array_equal(outs[k][1].cpu().numpy(),
                        ref[k][1]), (depth, k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(steps):
                pipe.submit_device(n, *ins[k % NB], outs[k % NB][0], None, outs[k % NB][1])
            pipe.wait()
            ms = (time.perf_counter() - t0) / steps * 1e3
            # join: a caller's stream sees the outputs of a submit that waited for that stream's inputs
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                x0 = ins[1][0].clone()
            g = torch.zeros(n, 12, dtype=torch.float64, device=dev)
            k = pipe.submit_device(n, x0, *ins[1][1:], g, after_stream=s.cuda_stream)
            pipe.join(s.cuda_stream, k)
            with torch.cuda.stream(s):
                gsum = g.clone()
            s.synchronize()
            assert np.array_equal(gsum.cpu().numpy(), ref[1][0])
        return ms
    ms1 = min(run(1, 24) for _ in range(2)); ms2 = min(run(2, 24) for _ in range(2))
    print(f"4096 x h10 first solves: {ms1:.3f} ms per batch alone, {ms2:.3f} ms with two in flight")
    assert ms2 < 0.95 * ms1, (ms1, ms2)

    # a warm-started population stays on its slot: slot 1 alone carries its own OSQP workspace from tick to tick
    sc = scen.config3_random_flat(nb=256, seed=9)
    cfgw = pkg.make_config(sc["params"], 10, warm_start=1)
    a = [t(sc["x0"]), t(sc["xref"]), t(sc["R"]), t(sc["foot"]), t(sc["contact"], torch.uint8)]
    with pkg.Engine(cfgw, 256, 0) as eng, pkg.Pipeline(cfgw, 256, 0, depth=2) as pipe:
        for tick in range(3):
            g0 = torch.zeros(256, 12, dtype=torch.float64, device=dev); i0 = torch.zeros(256, dtype=torch.int32, device=dev)
            g1 = torch.zeros(256, 12, dtype=torch.float64, device=dev); i1 = torch.zeros(256, dtype=torch.int32, device=dev)
            eng.solve_device(256, *a, g0, None, i0)
            assert pipe.submit_device(256, *a, g1, None, i1, slot=1, fresh=False) == 1
            pipe.wait(1); torch.cuda.synchronize()
            assert np.array_equal(g0.cpu().numpy(), g1.cpu().numpy()) and np.array_equal(i0.cpu().numpy(), i1.cpu().numpy()), tick
        assert i1.float().mean().item() < 40   # warm: 25 iterations for nearly every QP


def test_batch_pipeline_argument_errors(pkg, scen):
    import ctypes as C
    sc = scen.config3_random_flat(nb=8)
    cfg = pkg.make_config(sc["params"], 10, warm_start=0)
    lib = pkg.load_library()
    p = C.c_void_p()
    assert lib.a1mpc_pipeline_create(C.byref(cfg), 8, 0, 9, C.byref(p)) != 0 and not p          # depth > 8
    assert lib.a1mpc_pipeline_create(C.byref(cfg), 0, 0, 2, C.byref(p)) != 0
    bad = pkg.make_config(sc["params"], 7)
    assert lib.a1mpc_pipeline_create(C.byref(bad), 8, 0, 2, C.byref(p)) != 0 and not p          # unsupported horizon, nothing leaked
    with pkg.Pipeline(cfg, 8, 0, depth=0) as pipe:
        assert pipe.depth == 2   # default: two batches in flight at every size
        with pkg.Pipeline(cfg, 4096, 0, depth=0) as big:
            assert big.depth == 2
        with pytest.raises(pkg.A1MpcError):
            pipe.submit_device(8, None, None, None, None, None, None)
        with pytest.raises(pkg.A1MpcError):
            pipe.wait(5)
        pipe.wait()   # nothing submitted yet: returns at once
        assert lib.a1mpc_pipeline_wait(None, -1) != 0
    lib.a1mpc_pipeline_destroy(None)


def _oracle_update_ticks(oracle, pr, st, scs, carries):
    """one update-path tick of every robot b (its own carry) on the oracle"""
    n = len(scs["x0"])
    grf = np.zeros((n, 12)); it = np.zeros(n, np.int32); stt = np.zeros(n, np.int32)
    for b in range(n):
        o = oracle.mpc_solve_update(pr, st, scs["x0"][b], scs["xref"][b], scs["R"][b], scs["foot"][b], scs["contact"][b], carries[b])
        grf[b] = o["grf"]; it[b] = o["info"].iters; stt[b] = o["info"].status
    return grf, it, stt


@pytest.mark.parametrize("n,h", [(1, 10), (300, 10), (2600, 10), (8300, 10), (1300, 16), (1100, 20)])
def test_update_path_warm_start_2_matches_oracle(pkg, oracle, scen, n, h):
    """warm_start = 2: the reference's tick >= 2 UPDATE path on the latency kernel (n = 1), the fused kernel (300; since round 4 also the
    warm ticks of 2600 x h10),
    and the split pipeline's update-path instantiations -- set-up kernel + persistent rows at h = 10 (8300 > the 8192 up to which warm
    ticks run fused), the CU-wide
    kernel at h = 16 (1300), the one-wave kernel at h = 20 (1100) -- against the oracle's restatement of OSQP's update functions
    (orc_mpc_solve_update): every robot
    carries its own workspace through a sequence of slowly moving states with a contact switch; same iteration count and status on every QP
    of every tick, forces
    within the parity tolerance."""
    rng = np.random.default_rng(100 + n)
    sc = scen.config3_random_flat(nb=n, seed=900 + n, horizon=h)
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    carries = [oracle.update_carry(h) for _ in range(n)]
    ticks = 8 if n == 1 else (5 if n == 300 else 3)
    with _engine(pkg, sc, n, warm_start=2) as eng:
        for t in range(ticks):
            if t > 0:
                sc["x0"][:, :12] += rng.normal(0, 2e-3, (n, 12)); sc["foot"] += rng.normal(0, 1e-3, (n, 12))
            if t == 2:
                # every leg changes role: constraint types change, the carried z / y meet other bounds
                sc["contact"][:] = 1 - sc["contact"]
                sc["contact"][sc["contact"].sum(1) == 0] = [1, 0, 0, 1]
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
            grf, it, stt = _oracle_update_ticks(oracle, pr, st, sc, carries)
            assert (out["iters"] == it).all() and (out["status"] == stt).all(), (n, t, int((out["iters"] != it).sum()))
            assert np.abs(out["grf"] - grf).max() <= TOL_FORCE_N, (n, t, np.abs(out["grf"] - grf).max())
    # the first tick of a handle and the tick after a1mpc_reset_warm_start are cold solves whatever the mode
    with _engine(pkg, sc, n, warm_start=2) as eng, _engine(pkg, sc, n, warm_start=0) as cold:
        a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); c = cold.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"],
                sc["contact"])
        assert np.array_equal(a["grf"], c["grf"]) and np.array_equal(a["iters"], c["iters"])
        eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); eng.reset_warm_start()
        a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        assert np.array_equal(a["grf"], c["grf"]) and np.array_equal(a["iters"], c["iters"])


def test_pipeline_slots_carry_their_own_update_path_workspace(pkg, scen):
    """two robot fleets on the two slots of a pipeline with warm_start = 2: each slot carries its own OSQP workspace (x, y, rho AND the
    update path's scalings /
    gradient / z), so every tick of a fleet equals the tick of a lone handle that solved the same sequence -- bit for bit, while the two
    fleets' launches overlap."""
    import torch
    n = 2600
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    fleets = [scen.config3_random_flat(nb=n, seed=70 + f) for f in range(2)]
    cfg = pkg.make_config(fleets[0]["params"], 10, warm_start=2)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    with pkg.Engine(cfg, n, 0) as e0, pkg.Engine(cfg, n, 0) as e1, pkg.Pipeline(cfg, n, 0, depth=2) as pipe:
        for tick in range(4):
            ins = []
            for f in range(2):
                if tick > 0:
                    fleets[f]["x0"][:, :12] += rng.normal(0, 2e-3, (n, 12))
                ins.append([t(fleets[f]["x0"]), t(fleets[f]["xref"]), t(fleets[f]["R"]), t(fleets[f]["foot"]),
                        t(fleets[f]["contact"], torch.uint8)])
            outs = [(torch.zeros(n, 12, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(4)]
            assert pipe.submit_device(n, *ins[0], outs[0][0], None, outs[0][1], slot=0, fresh=False) == 0
            assert pipe.submit_device(n, *ins[1], outs[1][0], None, outs[1][1], slot=1, fresh=False) == 1
            e0.solve_device(n, *ins[0], outs[2][0], None, outs[2][1]); e1.solve_device(n, *ins[1], outs[3][0], None, outs[3][1])
            pipe.wait(); torch.cuda.synchronize()
            for f in range(2):
                assert np.array_equal(outs[f][0].cpu().numpy(), outs[2 + f][0].cpu().numpy()) and np.array_equal(outs[f][1].cpu().numpy(),
                        outs[2 + f][1].cpu().numpy()), (tick, f)
        assert outs[0][1].float().mean().item() < 40   # warm ticks


def test_host_pointer_pipeline_matches_the_synchronous_entry(pkg, scen):
    """a1mpc_pipeline_submit / _wait: host arrays in, host arrays out (the reference's side of the boundary, S/A1RobotControl.h:44), two or
    three batches in flight.
    The inputs are snapshotted before submit returns (they are overwritten right behind it here); every batch comes back bit-identical to
    a1mpc_solve_batch,
    u_full / iters / status included; a slot that is resubmitted first delivers its previous batch; n = 0 and ragged sizes work."""
    n, NB = 3000, 5   # beyond the resident rows: the split pipeline
    scs = [scen.config3_random_flat(nb=n, seed=800 + k) for k in range(NB)]
    cfg = pkg.make_config(scs[0]["params"], 10, warm_start=0)
    ref = []
    with pkg.Engine(cfg, n, 0) as eng:
        for s in scs:
            eng.set_schedule(True)
            ref.append(eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"], want_u=True))
    for depth in (2, 3):
        with pkg.Pipeline(cfg, n, 0, depth=depth) as pipe:
            outs = [dict(grf=np.full((n, 12), np.nan), u=np.full((n, 120), np.nan), iters=np.full(n, -1, np.int32),
                    status=np.full(n, -99, np.int32)) for _ in range(NB)]
            for k, s in enumerate(scs):
                ins = [np.array(s[f]) for f in ("x0", "xref", "R", "foot", "contact")]
                slot = pipe.submit(*ins, outs[k], fresh=True)   # round-robin: a slot that still holds batch k - depth delivers it first
                assert slot == k % depth
                for a in ins:
                    a[...] = 0            # the caller's arrays are free again as soon as submit returns
                if k >= depth:            # batch k - depth was delivered by this submit
                    j = k - depth
                    assert np.array_equal(outs[j]["grf"], ref[j]["grf"]) and np.array_equal(outs[j]["iters"], ref[j]["iters"]), (depth, j)
            pipe.wait()
            for k in range(NB):
                assert np.array_equal(outs[k]["grf"], ref[k]["grf"]) and np.array_equal(outs[k]["u"], ref[k]["u"]), (depth, k)
                assert np.array_equal(outs[k]["iters"], ref[k]["iters"]) and np.array_equal(outs[k]["status"], ref[k]["status"]), (depth, k)
            # ragged: fewer QPs than max_batch, outputs optional, and an empty batch
            m = 37
            o = dict(grf=np.zeros((m, 12)))
            pipe.submit(scs[1]["x0"][:m], scs[1]["xref"][:m], scs[1]["R"][:m], scs[1]["foot"][:m], scs[1]["contact"][:m], o, slot=0)
            pipe.wait(0)
            with pkg.Engine(cfg, m, 0) as small:
                r = small.solve(scs[1]["x0"][:m], scs[1]["xref"][:m], scs[1]["R"][:m], scs[1]["foot"][:m], scs[1]["contact"][:m])
            assert np.array_equal(o["grf"], r["grf"])
            e = dict(grf=np.zeros((0, 12)))
            pipe.submit(scs[1]["x0"][:0], scs[1]["xref"][:0], scs[1]["R"][:0], scs[1]["foot"][:0], scs[1]["contact"][:0], e, slot=1)
            pipe.wait()
            with pytest.raises(pkg.A1MpcError):
                pipe.submit(np.zeros((n + 1, 13)), np.zeros((n + 1, 130)), np.zeros((n + 1, 9)), np.zeros((n + 1, 12)),
                        np.zeros((n + 1, 4), np.uint8), dict(grf=np.zeros((n + 1, 12))))


def test_update_path_carry_is_dropped_by_ticks_that_do_not_refresh_it(pkg, oracle, scen):
    """warm_start = 2.  (i) Round 6: a general-path tick (per-step feet + a contact schedule) of a batch BEYOND the resident rows of the general path's fused kernel now
    follows the update path too (the fused kernel in several rounds; until round 6 such a tick ran warm_start = 1 semantics and dropped the carry): a sequence
    fast, fast, general, fast, general, fast of 2048 robots reports mode 2 on every tick and a sample of the robots equals the oracle's persistent solver chained through
    the same sequence.  (ii) ADVICE round 2: a stretch in warm_start = 1 (a1mpc_update_config 2 -> 1 -> 2) rewrites the carried (x, y, rho) but not the update path's
    carry; the tick that follows must NOT pair the stale scalings / gradient / z with the fresh iterates: it is a fresh set-up warm-started from (x, y, rho) -- exactly
    what a warm_start = 1 handle that saw the same sequence does, bit for bit."""
    n = 2048   # (> 1536 = the resident rows of the general path's fused kernel at h = 10)
    rng = np.random.default_rng(77)
    sc = scen.config3_random_flat(nb=n, seed=4242)
    seq = []
    for t in range(6):
        if t > 0:
            sc["x0"][:, :12] += rng.normal(0, 2e-3, (n, 12))
        seq.append({k: np.array(v) if isinstance(v, np.ndarray) else v for k, v in sc.items()})
    feet_steps = lambda s: np.repeat(s["foot"][:, None, :], 10, axis=1) + rng.normal(0, 1e-3, (n, 10, 12))
    f2 = feet_steps(seq[2]).reshape(n, 120); c2 = np.ascontiguousarray(np.repeat(seq[2]["contact"][:, None, :], 10, axis=1).reshape(n, 40))
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    checked = list(range(0, n, 97))
    carries = {b: oracle.update_carry(10) for b in checked}
    worst = 0.0
    with _engine(pkg, sc, n, warm_start=2) as e2:
        for t, gen in enumerate([0, 0, 1, 0, 1, 0]):
            q = seq[t]
            if gen:
                out = e2.solve_strided(q["x0"], q["xref"], q["R"], f2, 12, c2, 4)
            else:
                out = e2.solve(q["x0"], q["xref"], q["R"], q["foot"], q["contact"])
            assert e2.last_warm_start_mode() == 2, t
            for b in checked:
                o = oracle.mpc_solve_update(pr, st, q["x0"][b], q["xref"][b], q["R"][b], f2[b] if gen else q["foot"][b], c2[b] if gen else q["contact"][b], carries[b],
                                            foot_stride=12 if gen else 0, contact_stride=4 if gen else 0)
                assert out["iters"][b] == o["info"].iters and out["status"][b] == o["info"].status, (t, gen, b, out["iters"][b], o["info"].iters)
                worst = max(worst, np.abs(out["grf"][b] - o["grf"]).max())
    assert worst <= 1e-7, worst
    # the same through a1mpc_update_config: 2 -> 1 -> 2 leaves no stale carry behind
    with _engine(pkg, sc, n, warm_start=2) as e2, _engine(pkg, sc, n, warm_start=1) as e1:
        for t in (0, 1):
            e2.solve(seq[t]["x0"], seq[t]["xref"], seq[t]["R"], seq[t]["foot"], seq[t]["contact"])
        x, y, rho = e2.get_warm_start(n); e1.set_warm_start(x, y, rho)
        e2.update_config(pkg.make_config(sc["params"], 10, warm_start=1))
        for t in (2, 3):
            a = e2.solve(seq[t]["x0"], seq[t]["xref"], seq[t]["R"], seq[t]["foot"], seq[t]["contact"])
            b = e1.solve(seq[t]["x0"], seq[t]["xref"], seq[t]["R"], seq[t]["foot"], seq[t]["contact"])
            assert np.array_equal(a["grf"], b["grf"]) and np.array_equal(a["iters"], b["iters"])
        e2.update_config(pkg.make_config(sc["params"], 10, warm_start=2))
        a = e2.solve(seq[4]["x0"], seq[4]["xref"], seq[4]["R"], seq[4]["foot"], seq[4]["contact"])
        b = e1.solve(seq[4]["x0"], seq[4]["xref"], seq[4]["R"], seq[4]["foot"], seq[4]["contact"])
        assert np.array_equal(a["grf"], b["grf"]) and np.array_equal(a["iters"],
                b["iters"]), "stale update-path carry used after a1mpc_update_config"


def test_update_path_injected_warm_start_is_reexpressed_on_the_workspace(pkg, oracle, scen):
    """with warm_start = 2 an a1mpc_warm_start(x, y, rho) no longer clears the update path's carry; it does what
    osqp_warm_start_x / _y do on the
    reference's persistent solver -- x and y replace the iterates, z becomes A x, the previous tick's scalings / gradient / bounds stay --
    and the next tick follows
    the update path from there.  Checked against the oracle started from exactly that workspace (carry_from_workspace with the ENGINE's
    scalings of the last tick,
    the injected x and y, z = A x): same iteration count, forces within the parity tolerance, on every robot; and the tick differs from
    what a cleared carry gives."""
    n = 48
    rng = np.random.default_rng(91)
    sc = scen.config3_random_flat(nb=n, seed=9100)
    seq = []
    for t in range(4):
        if t > 0:
            sc["x0"][:, :12] += rng.normal(0, 2e-3, (n, 12))
        seq.append({k: np.array(v) if isinstance(v, np.ndarray) else v for k, v in sc.items()})
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1); h = 10; mu = sc["params"]["mu"]
    with _engine(pkg, sc, n, warm_start=2) as eng:
        for t in (0, 1, 2):
            eng.solve(seq[t]["x0"], seq[t]["xref"], seq[t]["R"], seq[t]["foot"], seq[t]["contact"])
        x, y, rho = eng.get_warm_start(n); D, E, c = eng.get_workspace_scaling(n)
        # the injected state: NOT what the last tick left
        xi = x * (1.0 + rng.normal(0, 0.02, x.shape)); yi = y * (1.0 + rng.normal(0, 0.02, y.shape)); ri = rho * 1.5
        eng.set_warm_start(xi, yi, ri)
        zi = eng.get_workspace_z(n)
        f = xi.reshape(n, h, 4, 3)
        zA = np.stack([f[..., 0] + mu * f[..., 2], f[..., 0] - mu * f[..., 2], f[..., 1] + mu * f[..., 2], f[..., 1] - mu * f[..., 2],
                f[..., 2]], axis=-1).reshape(n, 20 * h)
        assert np.abs(zi - zA).max() < 1e-12     # z = A x (osqp_warm_start_x)
        out = eng.solve(seq[3]["x0"], seq[3]["xref"], seq[3]["R"], seq[3]["foot"], seq[3]["contact"], want_u=True)
        assert eng.last_warm_start_mode() == 2
    worst = 0.0
    for i in range(n):
        # the previous tick's data stays in the workspace
        P, g, _, l, u, _ = oracle.mpc_form(pr, seq[2]["x0"][i], seq[2]["xref"][i], seq[2]["R"][i], seq[2]["foot"][i], seq[2]["contact"][i])
        carry = oracle.carry_from_workspace(h, xi[i], yi[i], zA[i], ri[i], D[i], E[i], c[i], P, g, l, u)
        r = oracle.mpc_solve_update(pr, st, seq[3]["x0"][i], seq[3]["xref"][i], seq[3]["R"][i], seq[3]["foot"][i], seq[3]["contact"][i],
                carry)
        assert out["iters"][i] == r["info"].iters and out["status"][i] == r["info"].status, (i, out["iters"][i], r["info"].iters)
        worst = max(worst, float(np.abs(out["u"][i] - r["u"]).max()))
    assert worst <= TOL_FORCE_N, worst
    # ... and this is not what a cleared carry (a fresh set-up warm-started from the same x, y, rho: mode 1) would have returned
    with _engine(pkg, sc, n, warm_start=1) as e1:
        e1.set_warm_start(xi, yi, ri)
        m1 = e1.solve(seq[3]["x0"], seq[3]["xref"], seq[3]["R"], seq[3]["foot"], seq[3]["contact"], want_u=True)
    assert np.abs(m1["u"] - out["u"]).max() > 1e-6
    print(f"injected warm start on the update path: {n} robots, worst |du| vs the oracle from the same workspace {worst:.2e} N")


@pytest.mark.parametrize("n", [1, 200])
def test_update_path_reinitialises_on_a_hessian_pattern_change(pkg, oracle, scen, n):
    """warm_start = 2 and the OsqpEigen branch SURVEY 8(c) names: when exact zeros of the reference's dense Hessian appear or vanish
    (fixture T's weights: level <->
    pitched), updateHessianMatrix re-initialises the solver (rho back to settings.rho, fresh scaling) and warm-starts it with the
    workspace's SCALED iterates
    (S/A1RobotControl.cpp:533-538, S/ConvexMpc.cpp:211).  The kernels find the change in the zero patterns of U and V, the oracle in the
    dense P: same ticks re-initialise,
    same iteration counts, forces within the parity tolerance.  Two re-initialisations only -- the oracle's own two linear-system back ends
    drift apart by 1000x per
    re-initialised solve on this ill-conditioned QP (tests/test_emu_parity.py)."""
    T = scen.scenario_T(); p = T["params"]; h = 10
    pr = oracle_params(oracle, T); st = oracle.default_settings(warm_start=1)
    rng = np.random.default_rng(12)
    dz = rng.uniform(-0.01, 0.01, n)
    carries = [oracle.update_carry(h) for _ in range(n)]
    nominal = np.array([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35]).reshape(4, 3)
    with _engine(pkg, T, n, warm_start=2) as eng:
        for t, pitch in enumerate([0.0, 0.0, 0.02, 0.03, 0.0, 0.0]):
            R = scen.rot_zyx(0.0, pitch, 0.0)
            foot = (R @ nominal.T).T.reshape(12) if pitch else nominal.reshape(12)
            x0 = np.tile(np.array([0.0, pitch, 0.0, 0.0, 0.0, 0.15 + 0.001 * t, 0, 0, 0, 0, 0, 0, -9.8]), (n, 1)); x0[:, 5] += dz
            xref = np.stack([oracle.mpc_reference(h, p["dt"], x0[b, 0:3], x0[b, 3:6], R.reshape(9), np.zeros(3), np.zeros(3), np.zeros(3),
                    0.15) for b in range(n)])
            sc = dict(x0=x0, xref=xref, R=np.tile(R.reshape(9), (n, 1)), foot=np.tile(foot, (n, 1)),
                    contact=np.tile(np.array([1, 0, 1, 0], np.uint8), (n, 1)))
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
            re = []
            for b in range(n):
                o = oracle.mpc_solve_update(pr, st, x0[b], xref[b], sc["R"][b], sc["foot"][b], sc["contact"][b], carries[b])
                re.append(o["info"].reinit)
                assert out["iters"][b] == o["info"].iters and out["status"][b] == o["info"].status, (n, t, b, out["iters"][b],
                        o["info"].iters)
                assert np.abs(out["grf"][b] - o["grf"]).max() <= TOL_FORCE_N, (n, t, b)
            assert all(r == (1 if t in (2, 4) else 0) for r in re), (t, re[:8])


# ------------------------------------------------------------------------------------------------------------ round 3: the big batches,
# gated
@pytest.mark.parametrize("gen,n,h", [("config5_divergent", 32768, 20), ("config3_random_flat", 65536, 10), ("config4_random_h16", 65536, 16)])
def test_full_size_batches_every_qp_vs_oracle(pkg, oracle, scen, gen, n, h):
    """BASELINE configs[4] as ONE launch of 32768 x h20 (mixed contact patterns, 0.5 rad pitch), BASELINE's upper batch, 65536 x h10, as one
    launch, and the whole of BASELINE configs[3], 65536 x h16, on one GPU -- what `bench.py --config 4` solves at N = 1 -- EVERY QP against
    the oracle (all host threads): same iteration count and status on every QP, forces within the parity tolerance."""
    sc = getattr(scen, gen)(nb=n)
    with _engine(pkg, sc, n, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=False)
    ref = oracle_batch(oracle, sc, want_u=False)
    r = compare(out, ref, min_same=1.0)
    assert (out["status"] == ref["status"]).all()
    print(f"{n} x h{h}:", r, "mean iterations", float(out["iters"].mean()))


@pytest.mark.parametrize("mode", [1, 2])
def test_ten_thousand_warm_started_ticks_batch_1(pkg, oracle, scen, mode):
    """VERDICT r2 item 6 / BASELINE configs[1]: 10 000 sequential warm-started trot ticks of ONE robot (the reference's operating point,
    S/A1RobotControl.cpp:522-538)
    through the host-pointer entry, in both warm-start semantics -- 1: fresh set-up + osqp_warm_start, 2: the reference's per-tick OSQP
    update path -- against the oracle
    chained the same way.  Mode 1: the same iteration count and status and forces within the parity tolerance on EVERY tick.
    Mode 2 asserts something on every tick as well.  The update path makes this tick sequence a chaotic map
    (independent 2 cm / 0.02 rad noise on
    every tick: each solve starts from iterates scaled for another problem), so last-bit differences between two implementations grow from
    tick to tick until they
    exceed the tolerance -- the oracle's OWN two back ends, Cholesky of the reduced system vs LDL' of the KKT matrix, are 0.5 N apart from
    tick 3354 on.  Between those
    (counted, rare) PARTINGS every tick is within the parity tolerance with the same iteration count.  ON a parting tick the engine's
    answer is checked on its own:
      (i)  OSQP's termination test (auxil.c check_termination: unscaled residuals against eps_abs + eps_rel x norms) evaluated on the
      ENGINE's (x, z, y) of that tick
           passes -- the engine stopped at a point OSQP itself accepts;
      (ii) the oracle is then RE-SEEDED from the engine's workspace (a1mpc_get_warm_start / _workspace_z / _workspace_scaling: everything
      the reference's persistent
           solver carries) instead of both sides restarting cold, and the next tick -- engine, double-precision oracle and the x87
           extended-precision build of the
           oracle (tests/x87.py), all three from that one state -- must agree tick-for-tick again: engine vs oracle within 1e-7 N with the
           same iteration count (a
           parting is accumulated chaos, not a per-tick discrepancy), and the engine no further from the extended-precision answer than the
           double oracle is (+ 1e-10 N)."""
    import x87
    nt = 10000
    sc = scen.config2_trot_sequence(nt)
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    h = sc["horizon"]
    wx = np.zeros(12 * h); wy = np.zeros(20 * h); rho = None
    carry = oracle.update_carry(h)
    errs = np.zeros(nt); its = 0; diverged = []; partings = []
    xpr = x87.params(sc["params"], h) if mode == 2 else None
    seeded = None     # mode 2: the workspace the oracle was re-seeded with at the previous (parting) tick
    with _engine(pkg, sc, 1, warm_start=mode) as eng:
        for t in range(nt):
            out = eng.solve(sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t])
            if mode == 1:
                r = oracle.mpc_solve(pr, st, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], warm_x=wx,
                        warm_y=wy, warm_rho=rho)
                wx, wy, rho = r["warm_x"], r["warm_y"], r["rho"]
            else:
                r = oracle.mpc_solve_update(pr, st, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], carry)
            same = out["iters"][0] == r["info"].iters and out["status"][0] == r["info"].status
            errs[t] = float(np.abs(out["grf"][0] - r["grf"]).max()); its += int(out["iters"][0])
            if mode == 1:
                assert same, (mode, t, out["iters"], r["info"].iters)
                continue
            if seeded is not None:   # (ii) the tick after a parting: all three from ONE state
                xr = x87.mpc_solve_update(xpr, x87.settings(warm_start=1), sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t],
                        sc["contact"][t], seeded)
                d_eng = float(np.abs(out["grf"][0] - xr["grf"]).max()); d_orc = float(np.abs(r["grf"] - xr["grf"]).max())
                partings[-1].update(next_tick=dict(iters=(int(out["iters"][0]), int(r["info"].iters), xr["iters"]),
                        engine_vs_oracle_N=errs[t], engine_vs_x87_N=d_eng, oracle_vs_x87_N=d_orc))
                assert same and errs[t] <= 1e-7, (t, partings[-1])
                # (observed: the engine is the CLOSER one on every parting, 1e-13 against 1e-11 N)
                assert d_eng <= d_orc + 1e-10, (t, partings[-1])
                seeded = None
            if not same or errs[t] > TOL_FORCE_N:
                assert eng.last_warm_start_mode() == 2
                ex, ey, erho = eng.get_warm_start(1); ez = eng.get_workspace_z(1); eD, eE, ec = eng.get_workspace_scaling(1)
                P, g, _, l, u, csr = oracle.mpc_form(pr, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t])
                k = 10.0 if out["status"][0] == 2 else 1.0     # (SOLVED_INACCURATE: OSQP's approximate test, 10 x the tolerances)
                ct = oracle.check_termination(P, g, csr, ex[0], ez[0], ey[0], eps_abs=k * st.eps_abs, eps_rel=k * st.eps_rel)
                # (i) a point OSQP's own termination test accepts
                assert out["status"][0] in (1, 2) and ct["ok"], (t, int(out["status"][0]), ct)
                partings.append(dict(tick=t, iters=(int(out["iters"][0]), int(r["info"].iters)), engine_vs_oracle_N=errs[t],
                        pri=(ct["pri_res"], ct["pri_tol"]), dua=(ct["dua_res"], ct["dua_tol"])))
                diverged.append(t); errs[t] = 0.0
                # the oracle goes on from the ENGINE's workspace
                carry = oracle.carry_from_workspace(h, ex[0], ey[0], ez[0], erho[0], eD[0], eE[0], ec[0], P, g, l, u)
                seeded = carry.copy()
    worst = float(errs.max())
    print(f"warm_start = {mode}: 10000 ticks, |dGRF| median {np.median(errs):.1e}, 99.9 % {np.quantile(errs, 0.999):.1e}, "
          f"worst {worst:.2e} N (tick {int(errs.argmax())}), "
          f"ticks above 1e-6 N: {int((errs > 1e-6).sum())}, "
          f"partings (oracle re-seeded from the engine's workspace): {diverged}, mean iterations {its / nt:.1f}")
    for pt in partings:
        print("  parting", pt)
    if mode == 1:    # fresh set-up + osqp_warm_start: the parity tolerance on every one of the 10 000 ticks
        assert worst <= TOL_FORCE_N, (mode, int(errs.argmax()), worst)
    # the update path: at most a handful of partings in 10 000 ticks, every other tick within the tolerance (by construction of the count)
    else:
        assert len(diverged) <= 10 and worst <= TOL_FORCE_N, (mode, diverged, int(errs.argmax()), worst)
        assert seeded is None or diverged[-1] == nt - 1


@pytest.mark.parametrize("gen,n", [("config3_random_flat", 4096), ("config4_random_h16", 2560), ("config5_divergent", 2048)])
def test_stage_cycles_through_the_abi(pkg, scen, gen, n):
    """VERDICT r3 item 9 / SURVEY 5 (the reference's t1..t6 stopwatches, S/A1RobotControl.cpp:491-553): a1mpc_set_profiling runs the
    clock-stamped instantiation of the
    persistent ADMM kernel -- bit-identical results -- and a1mpc_last_stage_cycles splits the solve stage into factor passes | iterations |
    residual checks."""
    sc = getattr(scen, gen)(nb=n)
    with _engine(pkg, sc, n, warm_start=0) as eng:
        a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        assert eng.last_stage_cycles()["qps"] == 0          # not profiled
        eng.set_profiling(True)
        b = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        cyc = eng.last_stage_cycles(); nf = eng.last_nfact(n)
        eng.set_profiling(False)
        c = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        assert eng.last_stage_cycles()["qps"] == 0
    for k in ("grf", "u", "iters", "status"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k
    assert cyc["qps"] == n and min(cyc["factor"], cyc["iterate"], cyc["check"]) > 0
    tot = cyc["factor"] + cyc["iterate"] + cyc["check"]
    per_it = cyc["iterate"] / float(a["iters"].sum()); per_f = cyc["factor"] / float(nf.sum())
    print(f"{gen} x {n}: factor {cyc['factor'] / tot:.3f} | iterate {cyc['iterate'] / tot:.3f} | "
          f"check {cyc['check'] / tot:.3f} of the solve stage; "
          f"{per_it:.0f} cycles per iteration, {per_f:.0f} per factor pass (wave-mates' stalls included)")
    assert 0.5 < cyc["iterate"] / tot < 0.95 and 0.03 < cyc["factor"] / tot < 0.45


@pytest.mark.parametrize("n,mode", [(4096, 1), (4096, 2), (1, 1), (1, 2), (200, 2)])
def test_tick_stage_cycles_of_the_fused_and_latency_kernels(pkg, scen, n, mode):
    """a1mpc_set_profiling + a1mpc_last_tick_stage_cycles on the ticks the reference actually runs -- warm-started ticks through the fused kernel
    (4096 robots) and the latency kernel (1 and 200 robots), both warm-start semantics: the clock-stamped instantiation gives the same bits as the plain one, every
    stage is filled, the stages add up to the whole tick, and a tick that is not profiled (or runs the split pipeline) reports qps = 0."""
    sc = scen.config3_random_flat(nb=max(n, 8))
    take_n = lambda k: sc[k][:n]
    rng = np.random.default_rng(31 + n)
    with _engine(pkg, sc, n, warm_start=mode) as eng:
        outs = []
        for t in range(5):
            x0 = take_n("x0").copy(); x0[:, :12] += rng.normal(0, 0.002, (n, 12)) * (t > 0)
            prof = t == 3
            eng.set_profiling(prof)
            o = eng.solve(x0, take_n("xref"), take_n("R"), take_n("foot"), take_n("contact"), want_u=True)
            cyc = eng.last_tick_stage_cycles()
            if prof:
                assert cyc["qps"] == n, cyc
                parts = sum(cyc[k] for k in eng.TICK_STAGES[:-1])
                assert all(cyc[k] > 0 for k in eng.TICK_STAGES) and abs(parts - cyc["total"]) <= 1e-9 * cyc["total"], cyc
                assert 0.15 < cyc["iterate"] / cyc["total"] < 0.8 and 0.1 < cyc["ruiz"] / cyc["total"] < 0.5, cyc
                print(f"{n} robots, mode {mode}:", {k: round(cyc[k] / cyc["total"], 3) for k in eng.TICK_STAGES[:-1]}, "cycles per QP", round(cyc["total"] / n))
            else:
                assert cyc["qps"] == 0, (t, cyc)     # tick 0 of 4096 robots runs the split pipeline (its stage record is a1mpc_last_stage_cycles'), the others are not profiled
            outs.append(o)
    # the same five ticks without ever touching the profiler: bit for bit
    rng = np.random.default_rng(31 + n)
    with _engine(pkg, sc, n, warm_start=mode) as eng:
        for t in range(5):
            x0 = take_n("x0").copy(); x0[:, :12] += rng.normal(0, 0.002, (n, 12)) * (t > 0)
            o = eng.solve(x0, take_n("xref"), take_n("R"), take_n("foot"), take_n("contact"), want_u=True)
            for k in ("grf", "u", "iters", "status"):
                assert np.array_equal(o[k], outs[t][k]), (t, k)


@pytest.mark.parametrize("var,n,ticks,warm,values", [("A1MPC_FUSED_QUEUE", 4096, 2, 0, "0,1"), ("A1MPC_WARM_ORDER", 4096, 4, 1, "0,1"), ("A1MPC_WARM_ORDER", 4096, 4, 2, "0,1"),
                                                    ("A1MPC_ZERO_COPY_MAX", 1, 4, 2, "0,8"), ("A1MPC_ZERO_COPY_MAX", 8, 3, 1, "0,8"), ("A1MPC_ZERO_COPY_MAX", 8, 2, 0, "0,8")])
def test_opt_in_scheduling_switches_change_nothing_but_the_schedule(pkg, var, n, ticks, warm, values):
    """Round 5's two measured-and-not-adopted trials stay in the library behind environment switches: A1MPC_FUSED_QUEUE=1 (the fused kernel as persistent wavefronts on
    the work queue, profiles/r05_fused_queue_trial.txt) and A1MPC_WARM_ORDER=1 (warm ticks launched in the order of the previous tick's costs,
    profiles/r05_warm_tick_order.txt).  Both only reorder independent QPs: forces, full solutions, iteration counts and statuses of every tick are bit-identical
    with the switch on and off (children of tools/env_ab.py).  Round 6: the same for the small-batch host path -- A1MPC_ZERO_COPY_MAX = 0 (inputs / outputs
    staged through device memory) against the default 8 (the kernels read and write the handle's pinned block over PCIe): batch 1 and batch 8, full solutions included,
    cold and both warm-start semantics; the batch-1 latency figures of bench.py and tests/cpp/latency_harness.cpp are measured on the zero-copy path."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "env_ab.py"), var, str(n), str(ticks), str(warm), values], capture_output=True, text=True, timeout=600)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 2 and all("digest" in x for x in rows), (r.stdout[-500:], r.stderr[-500:])
    assert rows[0]["digest"] == rows[1]["digest"] and rows[0]["solved"] == 1.0, rows
    print(var, [x["kernel_ms"] for x in rows])


def test_timing_events_can_be_turned_off(pkg, scen):
    """a1mpc_set_timing(h, 0): no HIP timing events around the launches (a 400 Hz loop does not read them) -- the same results, and the calls that read the events say so"""
    sc = scen.config3_random_flat(nb=64)
    with _engine(pkg, sc, 64, warm_start=0) as eng:
        a = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        assert eng.last_kernel_ms() > 0
        eng.set_timing(False)
        b = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        with pytest.raises(pkg.A1MpcError):
            eng.last_kernel_ms()
        eng.set_timing(True)
        c = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        assert eng.last_kernel_ms() > 0
    assert np.array_equal(a["grf"], b["grf"]) and np.array_equal(a["grf"], c["grf"]) and np.array_equal(a["iters"], b["iters"])


@pytest.mark.parametrize("mode,n,h", [(1, 4096, 10), (2, 4096, 10), (1, 1400, 16), (1, 1200, 20), (2, 1200, 20), (1, 2400, 20)])
def test_warm_ticks_of_a_large_batch_take_the_fused_kernel_and_match_the_oracle(pkg, oracle, scen, mode, n, h):
    """Round 4: second and later warm-started ticks of a batch size run the FUSED kernel up to 8192 QPs at h = 10 (solve_device_impl:
    nothing left for the queue to
    balance when every QP takes ~25 iterations), the first tick and any tick after a1mpc_set_schedule the split pipeline.  4096 robots,
    four ticks with slowly moving
    states, both warm-start semantics: every 16th robot is chained through the oracle the same way -- same iteration count and status,
    forces within the parity
    tolerance on every tick -- and a1mpc_last_stage_ms tells which pipeline ran (the fused kernel has no set-up stage of its own).  The h =
    16 and the larger h = 20 cases
    run the warm-start hand-off of the split pipeline (the CU-wide kernel at h = 16) the same way; h = 20 up to 2048 QPs takes the fused
    kernel's quads of rows."""
    ticks = 4
    rng = np.random.default_rng(404)
    sc = scen.config3_random_flat(nb=n, seed=4040, horizon=h)
    pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
    sub = np.arange(0, n, 16 if h == 10 else 8)
    wx = {int(i): np.zeros(12 * h) for i in sub}; wy = {int(i): np.zeros(20 * h) for i in sub}; rho = {int(i): None for i in sub}
    carry = {int(i): oracle.update_carry(h) for i in sub}
    staged = []
    with _engine(pkg, sc, n, warm_start=mode) as eng:
        for t in range(ticks):
            if t > 0:
                sc["x0"][:, :12] += rng.normal(0, 2e-3, (n, 12))
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
            staged.append(eng.last_stage_ms()[0] > 0.0)
            worst = 0.0
            for i in sub:
                i = int(i)
                if mode == 1:
                    r = oracle.mpc_solve(pr, st, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i], warm_x=wx[i],
                            warm_y=wy[i], warm_rho=rho[i])
                    wx[i], wy[i], rho[i] = r["warm_x"], r["warm_y"], r["rho"]
                else:
                    r = oracle.mpc_solve_update(pr, st, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i], carry[i])
                assert out["iters"][i] == r["info"].iters and out["status"][i] == r["info"].status, (mode, t, i, out["iters"][i],
                        r["info"].iters)
                worst = max(worst, float(np.abs(out["u"][i] - r["u"]).max()))
            assert worst <= TOL_FORCE_N, (mode, t, worst)
    if h == 10 or (h == 20 and n <= 2048):
        # tick 0: split pipeline (set-up stage timed); ticks 1..: the fused kernel (h = 20: its quad of rows, up to 2048 QPs)
        assert staged[0] and not any(staged[1:]), staged
    else:
        # h = 16, larger h = 20 batches: warm ticks stay on the split pipeline (the CU-wide kernel at h = 16), where the fused kernel loses
        assert all(staged), staged


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


@pytest.mark.parametrize("h,n", [(20, 2100), (16, 2600)])
def test_quad_of_rows_kernels_leave_the_twin_pairs_bits(pkg, h, n):
    """h = 20 (one QP per wavefront) and waves 1-3 of the CU-wide kernel at h = 16 run the four rows of a wavefront as a QUAD on one QP
    (RowSolver<.., QUAD>): the per-lane state
    split four ways, the chains untouched.  Against the twin-pair kernels of the same library (A1MPC_QUAD=0 in a child process): forces,
    the full solution, iteration counts and
    statuses of first solves, solves in history order and three warm-started ticks -- the same bits."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_quad.py"), pkg.build.LIB_PATH, str(h), str(n), "1"],
            capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["bit_identical"] is True, r.stdout[-800:]


def test_the_soak_tail_is_double_precisions_own_noise(pkg, oracle, scen):
    """The one QP of this round's 819 200-QP soak (tests/tools/soak_parity.py 20000 200 10, profiles/r04_parity_soak_1M.txt) where engine
    and oracle part by more than the
    1e-5 N bar: seed 20160, QP 0 -- 100 iterations, rho adapted down to 5e-4, same iteration count and status, 4.4e-5 N between the two.
    On this QP the double-precision
    oracle's own answer moves by up to 2.9e-5 N when one word of x0 moves by one ulp (the x87 build: 7e-9 N -- the QP is not ill-posed,
    double precision is noisy on it),
    so no two double-precision implementations of the iterate sequence can be held to 1e-5 N here: the engine has to stay within 3 x that
    band, and its neighbours in
    the batch within the usual bar."""
    sc = scen.config3_random_flat(nb=4096, seed=20160, param_set="gazebo")
    with _engine(pkg, sc, 4096, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
    pr = oracle_params(oracle, sc)
    base, med, mx = noise_band(oracle, pr, sc, 0)
    assert out["iters"][0] == base["info"].iters and out["status"][0] == base["info"].status
    d = np.abs(out["grf"][0] - base["grf"].ravel()).max()
    assert mx > TOL_FORCE_N, (mx, "the oracle's noise band on this QP used to exceed the parity bar: has the oracle's arithmetic changed?")
    assert d <= 3.0 * mx, (d, med, mx)
    ref = oracle_batch(oracle, take(sc, 256), want_u=False)
    same = out["iters"][1:256] == ref["iters"][1:256]
    assert same.all() and np.abs(out["grf"][1:256] - ref["grf"][1:256]).max() <= TOL_FORCE_N
