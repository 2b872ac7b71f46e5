"""GPU, through the C ABI: the round-6 pipeline entries -- a1mpc_pipeline_submit_strided_device / _submit_strided / _submit_ticks_device (per-step feet and contact schedules,
S/ConvexMpc.h:74 + S/test/test_mpc.cpp:106-122, and the compact tick records, S/A1RobotControl.cpp:452-488, with two batches in flight).  Every batch must come back
bit-identical to the lone handle's entry point, whose results the oracle checks."""
import numpy as np
import pytest

from helpers import TOL_FORCE_N

pytestmark = pytest.mark.gpu


def _strided_inputs(scen, rng, h, nb, seed):
    sc = scen.config3_random_flat(nb=nb, horizon=h, seed=seed)
    p = sc["params"]
    vd = rng.uniform(-0.6, 0.6, (nb, 1, 1, 3))
    foot = (sc["foot"].reshape(nb, 1, 4, 3) - vd * p["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(nb, h * 12)
    sw = rng.integers(0, h + 1, (nb, 4)); first = rng.integers(0, 2, (nb, 4))
    contact = np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :], 1 - first[:, None, :]).astype(np.uint8).reshape(nb, h * 4)
    return sc, np.ascontiguousarray(foot), np.ascontiguousarray(contact)


@pytest.mark.parametrize("h,n", [(10, 4096), (10, 600), (16, 1500), (20, 1100)])
def test_pipelined_general_path_is_the_lone_handle_bit_for_bit(pkg, oracle, scen, h, n):
    """device pointers and host arrays, batches beyond and within the general path's resident rows; batch 0 vs the oracle's strided formation on a sample"""
    import torch
    NB = 3
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(600 + h)
    cases = [_strided_inputs(scen, rng, h, n, 700 + 10 * h + k) for k in range(NB)]
    cfg = pkg.make_config(cases[0][0]["params"], h, warm_start=0)
    ref = []
    with pkg.Engine(cfg, n, 0) as eng:
        for sc, foot, contact in cases:
            eng.set_schedule(True)
            ref.append(eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4, want_u=True))
    pr = oracle.mpc_params(h, **{k: cases[0][0]["params"][k] for k in ("dt", "mu", "fz_min", "fz_max", "q", "r", "mass", "inertia")}); st = oracle.default_settings()
    sc, foot, contact = cases[0]
    for b in range(0, n, max(1, n // 12)):
        r = oracle.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=12, contact_stride=4)
        assert ref[0]["iters"][b] == r["info"].iters and ref[0]["status"][b] == r["info"].status
        assert np.abs(ref[0]["u"][b] - r["u"]).max() <= TOL_FORCE_N
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    ins = [[t(sc["x0"]), t(sc["xref"]), t(sc["R"]), t(foot), t(contact, torch.uint8)] for sc, foot, contact in cases]
    with pkg.Pipeline(cfg, n, 0, depth=2) as pipe:
        # device pointers: five submits over three batches, two in flight
        outs = [(torch.zeros(n, 12, dtype=torch.float64, device=dev), torch.zeros(n, 12 * h, dtype=torch.float64, device=dev),
                 torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
        slots = []
        for k in range(5):
            x0, xr, R, ft, ct = ins[k % NB]; o = outs[k % NB]
            if k >= NB:
                pipe.wait()   # the output set of batch k % NB is handed to a new submit only once its previous batch has been waited for
            slots.append(pipe.submit_strided_device(n, x0, xr, R, ft, 12, ct, 4, o[0], o[1], o[2], o[3]))
        assert slots[:2] == [0, 1]
        pipe.wait()
        for k in range(NB):
            assert np.array_equal(outs[k][0].cpu().numpy(), ref[k]["grf"]) and np.array_equal(outs[k][1].cpu().numpy(), ref[k]["u"]), k
            assert np.array_equal(outs[k][2].cpu().numpy(), ref[k]["iters"]) and np.array_equal(outs[k][3].cpu().numpy(), ref[k]["status"]), k
        # host arrays in / out: the slot's pinned mirror hands each batch to its own arrays
        houts = [dict(grf=np.zeros((n, 12)), u=np.zeros((n, 12 * h)), iters=np.zeros(n, np.int32), status=np.zeros(n, np.int32)) for _ in range(NB)]
        for k in range(NB):
            sc, foot, contact = cases[k]
            pipe.submit_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4, houts[k])
        pipe.wait()
        for k in range(NB):
            for key in ("grf", "u", "iters", "status"):
                assert np.array_equal(houts[k][key], ref[k][key]), (k, key)
        # (0, 0, NULL) is the plain submit: the fast kernels, the plain entry's bits
        sc = cases[0][0]
        plain = dict(grf=np.zeros((n, 12)), iters=np.zeros(n, np.int32))
        via = dict(grf=np.zeros((n, 12)), iters=np.zeros(n, np.int32))
        pipe.submit(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], plain); pipe.wait()
        pipe.submit_strided(sc["x0"], sc["xref"], sc["R"], sc["foot"], 0, sc["contact"], 0, via); pipe.wait()
        assert np.array_equal(plain["grf"], via["grf"]) and np.array_equal(plain["iters"], via["iters"])
        with pytest.raises(pkg.A1MpcError):
            pipe.submit_strided_device(n, ins[0][0], ins[0][1], ins[0][2], ins[0][3], 7, ins[0][4], 4, outs[0][0])   # foot_stride must be 0 or 12


def test_pipelined_tick_records_are_the_lone_handle_bit_for_bit(pkg, oracle, scen):
    """a1mpc_pipeline_submit_ticks_device == a1mpc_solve_batch_ticks (N1: x0 / x_ref built on the device from the 22-number record), two batches in flight"""
    import torch
    n, NB, h = 4096, 3, 10
    dev = torch.device("cuda:0")
    scs = [scen.config3_random_flat(nb=n, seed=900 + k) for k in range(NB)]
    cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
    ref = []
    with pkg.Engine(cfg, n, 0) as eng:
        for sc in scs:
            eng.set_schedule(True)
            ref.append(eng.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"]))
        eng.set_schedule(True)
        full = eng.solve(scs[0]["x0"], scs[0]["xref"], scs[0]["R"], scs[0]["foot"], scs[0]["contact"])
    assert np.array_equal(full["iters"], ref[0]["iters"]) and np.abs(full["grf"] - ref[0]["grf"]).max() < TOL_FORCE_N
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    ins = [[t(sc["tick"]), t(sc["R"]), t(sc["foot"]), t(sc["contact"], torch.uint8)] for sc in scs]
    outs = [(torch.zeros(n, 12, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(NB)]
    with pkg.Pipeline(cfg, n, 0, depth=2) as pipe:
        for k in range(NB):
            pipe.submit_ticks_device(n, *ins[k], outs[k][0], None, outs[k][1], outs[k][2])
        pipe.wait()
    for k in range(NB):
        assert np.array_equal(outs[k][0].cpu().numpy(), ref[k]["grf"]) and np.array_equal(outs[k][1].cpu().numpy(), ref[k]["iters"]) and np.array_equal(outs[k][2].cpu().numpy(), ref[k]["status"]), k
