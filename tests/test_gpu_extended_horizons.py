"""-m gpu: the extended horizons (4, 6, 8, 12, 14 -- SURVEY 8 a1: PLAN_HORIZON is a compile-time constant of the reference, S/A1Params.h:26, a run-time value of the
C ABI) through the C ABI against the oracle: the fast path's kernel family as it instantiates for them (latency kernel, fused kernel, set-up kernel + persistent rows),
cold, warm-started, on the update path, with a contact schedule and from tick records.  Per-step feet at these horizons: tests/test_gpu_general_path.py."""
import numpy as np
import pytest

from gpu_common import *  # noqa: F401,F403
from gpu_common import _engine, _oracle_update_ticks, _strided_inputs  # noqa: F401
from helpers import TOL_FORCE_N, compare, exact_resolver, oracle_batch, oracle_params, take

pytestmark = pytest.mark.gpu

EXTENDED = (4, 6, 8, 12, 14)


@pytest.mark.parametrize("h", EXTENDED)
def test_cold_solves_on_every_pipeline(pkg, oracle, scen, h):
    """one QP and 40 (latency kernel), 600 (fused kernel) and 3000 (set-up kernel + persistent rows: more QPs than resident rows) first solves -- every QP against the
    oracle: same iteration count and status, forces of every horizon step within the parity tolerance; the three pipelines agree bit for bit on the QPs they share"""
    sc = scen.config3_random_flat(nb=3000, seed=5200 + h, horizon=h)
    ref = oracle_batch(oracle, sc)
    outs = {}
    with _engine(pkg, sc, 3000, warm_start=0) as eng:
        for n in (1, 40, 600, 3000):
            s = take(sc, n)
            outs[n] = eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"], want_u=True)
            r = compare(outs[n], {k: v[:n] for k, v in ref.items() if v is not None}, min_same=1.0)
            if n == 3000:
                print("h", h, r, "iterations", dict(zip(*np.unique(outs[n]["iters"], return_counts=True))))
        assert eng.last_stage_ms()[0] > 0.0   # the 3000-QP solve went through the split pipeline (it has a set-up stage of its own)
    for n in (1, 40, 600):
        assert np.array_equal(outs[n]["u"], outs[3000]["u"][:n]) and np.array_equal(outs[n]["iters"], outs[3000]["iters"][:n]), n


@pytest.mark.parametrize("h", EXTENDED)
def test_other_scenarios_and_parameter_sets(pkg, oracle, scen, h):
    """the divergent configuration (all contact patterns, 0.5 rad pitch: forces on the pyramid's faces, S/BASELINE configs[4]) and the hardware / isaac weights"""
    sc = scen.config5_divergent(nb=256, horizon=h)
    with _engine(pkg, sc, 256, warm_start=0) as eng:
        out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
    compare(out, oracle_batch(oracle, sc), resolve=exact_resolver(oracle, sc))
    for ps in ("hardware", "isaac"):
        sc = scen.config3_random_flat(nb=48, param_set=ps, horizon=h)
        with _engine(pkg, sc, 48, warm_start=0) as eng:
            out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        compare(out, oracle_batch(oracle, sc), resolve=exact_resolver(oracle, sc))


@pytest.mark.parametrize("h,n", [(4, 1), (4, 300), (6, 1), (8, 2600), (12, 300), (14, 1), (12, 2600), (14, 300)])
def test_update_path_and_warm_start(pkg, oracle, scen, h, n):
    """warm_start = 2 (the reference's tick >= 2 update path, S/A1RobotControl.cpp:533-538) and warm_start = 1 over a sequence of slowly moving states with a contact
    switch, every robot with its own workspace: same iteration count and status on every QP of every tick as the oracle's restatement of OSQP's update functions"""
    for mode in (2, 1):
        rng = np.random.default_rng(300 + 7 * h + n)
        sc = scen.config3_random_flat(nb=n, seed=1900 + h + n, horizon=h)
        pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
        carries = [oracle.update_carry(h) for _ in range(n)]
        wx = [np.zeros(12 * h) for _ in range(n)]; wy = [np.zeros(20 * h) for _ in range(n)]; rho = [None] * n   # warm_start = 1: what the oracle's solver carries per robot
        with _engine(pkg, sc, n, warm_start=mode) as eng:
            for t in range(4 if n > 1000 else 6):
                if t > 0:
                    sc["x0"][:, :12] += rng.normal(0, 2e-3, (n, 12)); sc["foot"] += rng.normal(0, 1e-3, (n, 12))
                if t == 2:
                    sc["contact"][:] = 1 - sc["contact"]
                    sc["contact"][sc["contact"].sum(1) == 0] = [1, 0, 0, 1]
                out = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
                if mode == 2:
                    grf, it, stt = _oracle_update_ticks(oracle, pr, st, sc, carries)
                else:
                    grf = np.zeros((n, 12)); it = np.zeros(n, np.int32); stt = np.zeros(n, np.int32)
                    for b in range(n):
                        o = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b], warm_x=wx[b], warm_y=wy[b], warm_rho=rho[b])
                        grf[b] = o["grf"]; it[b] = o["info"].iters; stt[b] = o["info"].status
                        wx[b], wy[b], rho[b] = o["warm_x"], o["warm_y"], o["rho"]
                assert (out["iters"] == it).all() and (out["status"] == stt).all(), (mode, h, n, t, int((out["iters"] != it).sum()))
                assert np.abs(out["grf"] - grf).max() <= TOL_FORCE_N, (mode, h, n, t, np.abs(out["grf"] - grf).max())
            if mode == 2:
                assert eng.last_warm_start_mode() == 2


@pytest.mark.parametrize("h,nb", [(4, 64), (6, 700), (8, 1), (12, 4500), (14, 64)])
def test_contact_schedule_and_tick_records(pkg, oracle, scen, h, nb):
    """a per-step contact schedule with step-invariant feet (contact_stride = 4) runs the fast kernels at every horizon a1mpc_create accepts (a1mpc.h); tick records
     build the same x0 / x_ref as the caller would; per-step feet equal to the broadcast ones give the fast path's forces on the general kernels"""
    rng = np.random.default_rng(4100 + h + nb)
    sc, foot, fs, contact, cs = _strided_inputs(scen, rng, h, nb, False, True)
    pr = oracle_params(oracle, sc); st = oracle.default_settings()
    with _engine(pkg, sc, nb, warm_start=0) as eng:
        out = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], sc["foot"], 0, contact, 4, want_u=True)
        worst = 0.0
        for b in range(0, nb, max(1, nb // 48)):
            r = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], contact[b], foot_stride=0, contact_stride=4)
            assert out["iters"][b] == r["info"].iters and out["status"][b] == r["info"].status, (b, out["iters"][b], r["info"].iters)
            worst = max(worst, np.abs(out["u"][b] - r["u"]).max(), np.abs(out["grf"][b] - r["grf"]).max())
        assert worst <= TOL_FORCE_N, worst
        u = out["u"].reshape(nb, h, 4, 3); c = contact.reshape(nb, h, 4)
        assert np.abs(u[c == 0]).max() < 1.0   # a leg in swing at step t carries no force at step t
        a = eng.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        b_ = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        assert (a["iters"] == b_["iters"]).all() and np.abs(a["u"] - b_["u"]).max() < TOL_FORCE_N
        feet = np.ascontiguousarray(np.tile(sc["foot"], (1, h)))
        gen = eng.solve_strided(sc["x0"], sc["xref"], sc["R"], feet, 12, contact, 4, want_u=True)   # the same QPs through the general kernels (per-step feet that happen to be equal)
        assert np.array_equal(gen["iters"], out["iters"]) and np.abs(gen["u"] - out["u"]).max() <= 1e-7
        again = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)
        assert np.array_equal(again["u"], b_["u"])


def test_pipeline_and_unsupported_horizons(pkg, oracle, scen):
    """two batches in flight at an extended horizon leave the lone handle's bits; odd horizons, 2 and 18 are refused at a1mpc_create"""
    h, n = 12, 2600
    scs = [scen.config3_random_flat(nb=n, seed=6100 + k, horizon=h) for k in range(3)]
    cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
    with pkg.Engine(cfg, n, 0) as eng:
        lone = [eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"]) for s in scs]
    outs = [dict(grf=np.zeros((n, 12)), iters=np.zeros(n, np.int32), status=np.zeros(n, np.int32)) for _ in scs]
    with pkg.Pipeline(cfg, n, 0, depth=2) as pipe:
        for s, o in zip(scs, outs):
            pipe.submit(s["x0"], s["xref"], s["R"], s["foot"], s["contact"], o)
        pipe.wait()
    for a, b in zip(lone, outs):
        assert np.array_equal(a["grf"], b["grf"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["status"], b["status"])
    for bad in (2, 3, 5, 7, 18, 22, 0, -1):
        with pytest.raises(pkg.A1MpcError):
            pkg.Engine(pkg.make_config(scs[0]["params"], bad), 4, 0)
