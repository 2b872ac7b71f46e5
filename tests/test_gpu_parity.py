"""-m gpu: the HIP path, called through the C ABI (liba1mpc.so), against the oracle on identical inputs."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import *  # noqa: F401,F403
from gpu_common import _engine, _oracle_update_ticks, _strided_inputs  # noqa: F401
from helpers import TOL_FORCE_BALANCE_N, TOL_FORCE_N, compare, exact_resolver, noise_band, oracle_batch, oracle_params, take

pytestmark = pytest.mark.gpu


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


def test_small_balance_batches_take_the_pinned_block(pkg, oracle, scen):
    """Round 6: a1mpc_balance_solve_batch with a handful of QPs (the drop-in's compute_grf with stance_leg_control_type = 0 is n = 1) reads and writes the handle's pinned block and
    polls the output words; larger batches keep the staged copies.  n = 1, 4, 8 against the first rows of an n = 9 call bit for bit, every QP at the oracle's iteration count."""
    sc = scen.balance_random(9)
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    qp, st = oracle.default_qp_params(), oracle.default_settings()
    with pkg.Engine(cfg, 16, 0) as eng:
        big = eng.balance_solve(sc["root_acc"], sc["R"], sc["Rz"], sc["foot"], sc["contact"])            # staged
        for b in range(9):
            r = oracle.balance_solve(qp, st, sc["root_acc"][b], sc["R"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
            assert big["iters"][b] == r["info"].iters and big["status"][b] == r["info"].status and np.abs(big["grf"][b] - r["grf"]).max() < TOL_FORCE_BALANCE_N
        for n in (1, 4, 8):
            for rep in range(3):   # (the block is reused call after call)
                a = eng.balance_solve(sc["root_acc"][:n], sc["R"][:n], sc["Rz"][:n], sc["foot"][:n], sc["contact"][:n])   # the pinned block, polled
                for k in ("grf", "f_world", "iters", "status"):
                    assert np.array_equal(a[k], big[k][:n]), (n, rep, k)


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


def test_small_host_batches_of_tick_records_take_the_pinned_block(pkg, oracle, scen):
    """Round 6: a1mpc_solve_batch_ticks with a handful of robots (the drop-in's compute_grf is n = 1) lets the kernel read and write the handle's pinned block and polls the
    output words, like a1mpc_solve_batch; larger batches keep the staged copies.  The same robots through both (n = 1, 3, 8 against the first rows of an n = 9 call; cold
    solves, so nothing depends on history), with and without the inputs' copy out: bit for bit, every QP at the oracle's iteration count -- and 300 warm-started
    batch-1 ticks of one robot equal the same ticks taken from the device-pointer entry."""
    sc = scen.config3_random_flat(nb=9, seed=4242)
    with _engine(pkg, sc, 16, warm_start=0) as eng:
        big = eng.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"], want_u=True)           # n = 9: staged
        compare(big, oracle_batch(oracle, sc), min_same=1.0)
        for n in (1, 3, 8):
            for want_u in (True, False):
                s = take(sc, n)
                a = eng.solve_ticks(sc["tick"][:n], s["R"], s["foot"], s["contact"], want_u=want_u)   # n <= 8: the pinned block, polled
                assert np.array_equal(a["grf"], big["grf"][:n]) and np.array_equal(a["iters"], big["iters"][:n]) and np.array_equal(a["status"], big["status"][:n]), (n, want_u)
                if want_u: assert np.array_equal(a["u"], big["u"][:n]), n
                b = eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"], want_u=want_u)      # the x0 / x_ref entry's small-batch path
                assert (a["iters"] == b["iters"]).all() and np.abs(a["grf"] - b["grf"]).max() < TOL_FORCE_N
    seq = scen.config2_trot_sequence(300)
    import torch
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tick = np.zeros((300, 22)); tick[:, 0:12] = seq["x0"][:, 0:12]; tick[:, 12:15] = seq["x0"][:, 0:3]; tick[:, 15] = 0.2; tick[:, 21] = 0.3
    cfg = pkg.make_config(seq["params"], 10, warm_start=1)
    with pkg.Engine(cfg, 4, 0) as e1, pkg.Engine(cfg, 4, 0) as e2:
        g = torch.zeros(1, 12, dtype=torch.float64, device=dev); it = torch.zeros(1, dtype=torch.int32, device=dev); stt = torch.zeros(1, dtype=torch.int32, device=dev)
        for t in range(300):
            a = e1.solve_ticks(tick[t], seq["R"][t], seq["foot"][t], seq["contact"][t])
            ins = [T(tick[t:t + 1]), T(seq["R"][t:t + 1]), T(seq["foot"][t:t + 1]), T(seq["contact"][t:t + 1].astype(np.uint8))]
            rc = e2.lib.a1mpc_solve_batch_ticks_device(e2._h, 1, *[C.c_void_p(x.data_ptr()) for x in ins], C.c_void_p(g.data_ptr()), None, C.c_void_p(it.data_ptr()), C.c_void_p(stt.data_ptr()), None)
            assert rc == 0
            nf = (C.c_int32 * 1)(); assert e2.lib.a1mpc_last_nfact(e2._h, 1, nf) == 0   # (synchronises the handle's stream, which the launch went to)
            assert np.array_equal(a["grf"][0], g.cpu().numpy()[0]) and a["iters"][0] == int(it[0]) and a["status"][0] == int(stt[0]), t


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


def test_horizon_1_mpc(pkg, oracle, scen):
    """the MPC path at horizon 1 (n = 12, m = 20), the shape BASELINE's north star pairs with the balance QP"""
    sc = scen.config3_random_flat(nb=64, horizon=1)
    with _engine(pkg, sc, 64, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    compare(out, oracle_batch(oracle, sc), min_same=1.0)


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
