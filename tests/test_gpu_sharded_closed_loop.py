"""GPU, through the C ABI: the round-6 entries of the sharded handle -- tick records, device-resident batches, per-shard warm starts (cfg.warm_start = 1 and 2).  The
test box has ONE GPU: two / three shards on device 0 (transport 0: the peer copies degenerate to copies inside the device) and the RCCL transport with one rank.  Bar:
bit-identical to ONE engine handle ticking the same robots -- whose ticks the oracle checks in tests/test_gpu_update_path.py -- shard by shard (a shard's handle sees its slice
of the batch as batch positions 0 .. c-1, exactly like a lone handle given that slice)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ticks(scen, n, nticks, seed=77, horizon=10):
    """`nticks` consecutive ticks of the same n robots: slowly moving states, tick records and (x0, x_ref) of the same ticks"""
    base = scen.config3_random_flat(nb=n, seed=seed, horizon=horizon)
    rng = np.random.default_rng(seed)
    out = []
    x0 = base["x0"].copy(); tick = base["tick"].copy()
    for t in range(nticks):
        sc = dict(base)
        d = rng.normal(0, 2e-3, (n, 12))
        x0 = x0.copy(); x0[:, :12] += d
        tick = tick.copy(); tick[:, :12] += d
        sc["x0"] = x0; sc["tick"] = tick
        out.append(sc)
    return out


def _lone_reference(pkg, cfg, seq, n, splits, use_ticks):
    """the same ticks through one engine handle per shard range (what the sharded handle is built from)"""
    res = [dict(grf=np.zeros((n, 12)), iters=np.zeros(n, np.int32), status=np.zeros(n, np.int32)) for _ in seq]
    for (s0, c) in splits:
        if c == 0:
            continue
        with pkg.Engine(cfg, c, 0) as eng:
            for t, sc in enumerate(seq):
                sl = slice(s0, s0 + c)
                if use_ticks:
                    o = eng.solve_ticks(sc["tick"][sl], sc["R"][sl], sc["foot"][sl], sc["contact"][sl])
                else:
                    o = eng.solve(sc["x0"][sl], sc["xref"][sl], sc["R"][sl], sc["foot"][sl], sc["contact"][sl])
                for k in ("grf", "iters", "status"):
                    res[t][k][sl] = o[k]
    return res


def _ranges(n, G):
    base, rem = divmod(n, G)
    return [(g * base + min(g, rem), base + (1 if g < rem else 0)) for g in range(G)]


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("n,devs,h", [(601, [0, 0], 10), (3001, [0, 0, 0], 10), (601, [0, 0], 8)])   # (h = 8: one of the extended horizons)
def test_sharded_closed_loop_equals_lone_handles(pkg, scen, mode, n, devs, h):
    import torch
    seq = _ticks(scen, n, 4, horizon=h)
    cfg = pkg.make_config(seq[0]["params"], h, warm_start=mode)
    splits = _ranges(n, len(devs))
    ref_x = _lone_reference(pkg, cfg, seq, n, splits, use_ticks=False)
    ref_t = _lone_reference(pkg, cfg, seq, n, splits, use_ticks=True)
    assert all(int(r["iters"].max()) > 0 for r in ref_x)
    assert float(np.mean(ref_x[-1]["iters"])) < float(np.mean(ref_x[0]["iters"]))   # the later ticks ARE warm-started
    dev = torch.device("cuda:0")
    t_ = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    with pkg.ShardedEngine(cfg, n, devices=devs, transport=0) as sh:
        # host arrays, (x0, x_ref)
        for t, sc in enumerate(seq):
            o = sh.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
            for k in ("grf", "iters", "status"):
                assert np.array_equal(o[k], ref_x[t][k]), ("host x0/xref", t, k)
        tr = sh.last_transfer()
        moved = n   # transport 0 with host arrays: every shard is fed from the pinned mirror
        assert tr["scatter_bytes"] == moved * ((13 + 13 * h + 9 + 12) * 8 + 4) and tr["gather_bytes"] == moved * (96 + 8)
        # host arrays, tick records (a fresh closed loop: the shards' warm starts are dropped first)
        sh.reset_warm_start()
        for t, sc in enumerate(seq):
            o = sh.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"])
            for k in ("grf", "iters", "status"):
                assert np.array_equal(o[k], ref_t[t][k]), ("host ticks", t, k)
        # the batch resident on the root GPU, (x0, x_ref) and tick records
        grf = torch.zeros(n, 12, dtype=torch.float64, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
        stream = torch.cuda.Stream(device=dev)
        for form, ref in (("x", ref_x), ("t", ref_t)):
            sh.reset_warm_start()
            for t, sc in enumerate(seq):
                with torch.cuda.stream(stream):
                    ins = [t_(sc["x0"]), t_(sc["xref"])] if form == "x" else [t_(sc["tick"])]
                    ins += [t_(sc["R"]), t_(sc["foot"]), t_(sc["contact"], torch.uint8)]
                if form == "x":
                    sh.solve_device(n, *ins, grf, it, st, stream=stream.cuda_stream)
                else:
                    sh.solve_ticks_device(n, *ins, grf, it, st, stream=stream.cuda_stream)
                assert np.array_equal(grf.cpu().numpy(), ref[t]["grf"]) and np.array_equal(it.cpu().numpy(), ref[t]["iters"]) and np.array_equal(st.cpu().numpy(), ref[t]["status"]), (form, t)
            tr = sh.last_transfer()
            moved = n - splits[0][1]   # the root's own shard is solved in place
            per_qp = ((13 + 13 * h) if form == "x" else 22) * 8 + (9 + 12) * 8 + 4
            assert tr["scatter_bytes"] == moved * per_qp and tr["gather_bytes"] == moved * (96 + 8), (form, tr)


def test_sharded_rccl_transport_one_rank_device_forms(pkg, scen):
    """transport 1 on the one device of the test box: communicator set-up, the root solving in place on the caller's arrays (no peer to send to)"""
    import torch
    n = 512
    seq = _ticks(scen, n, 3, seed=5)
    cfg = pkg.make_config(seq[0]["params"], 10, warm_start=2)
    ref = _lone_reference(pkg, cfg, seq, n, [(0, n)], use_ticks=True)
    dev = torch.device("cuda:0")
    t_ = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    grf = torch.zeros(n, 12, dtype=torch.float64, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev)
    with pkg.ShardedEngine(cfg, n, devices=[0], transport=1) as sh:
        for t, sc in enumerate(seq):
            sh.solve_ticks_device(n, t_(sc["tick"]), t_(sc["R"]), t_(sc["foot"]), t_(sc["contact"], torch.uint8), grf, it, None)
            torch.cuda.synchronize()
            assert np.array_equal(grf.cpu().numpy(), ref[t]["grf"]) and np.array_equal(it.cpu().numpy(), ref[t]["iters"]), t
        assert sh.last_transfer() == dict(scatter_bytes=0, gather_bytes=0)
        sh.reset_warm_start()
        for t, sc in enumerate(seq):
            o = sh.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"])
            assert np.array_equal(o["grf"], ref[t]["grf"]) and np.array_equal(o["iters"], ref[t]["iters"]), t
        with pytest.raises(pkg.A1MpcError):
            sh.solve_ticks_device(n, None, t_(seq[0]["R"]), t_(seq[0]["foot"]), t_(seq[0]["contact"], torch.uint8), grf)


def test_sharded_and_pipeline_argument_errors(pkg, scen):
    """the new entries refuse what the old ones refuse: null pointers, a batch beyond max_batch, tick records at horizon 1, bad strides -- with a status, never a crash"""
    import ctypes as C
    import torch
    n = 64
    sc = scen.config3_random_flat(nb=n)
    lib = pkg.load_library()
    dev = torch.device("cuda:0")
    t_ = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    cfg = pkg.make_config(sc["params"], 10, warm_start=1)
    with pkg.ShardedEngine(cfg, n, devices=[0, 0], transport=0) as sh:
        grf = torch.zeros(n, 12, dtype=torch.float64, device=dev)
        with pytest.raises(pkg.A1MpcError):
            sh.solve_ticks_device(n + 1, t_(sc["tick"]), t_(sc["R"]), t_(sc["foot"]), t_(sc["contact"], torch.uint8), grf)     # n > max_batch
        with pytest.raises(pkg.A1MpcError):
            sh.solve_device(n, None, t_(sc["xref"]), t_(sc["R"]), t_(sc["foot"]), t_(sc["contact"], torch.uint8), grf)          # null x0
        with pytest.raises(pkg.A1MpcError):
            sh.solve_ticks_device(n, t_(sc["tick"]), t_(sc["R"]), t_(sc["foot"]), t_(sc["contact"], torch.uint8), None)         # null output
        h = C.c_void_p()
        assert lib.a1mpc_sharded_handle(sh._h, 2, C.byref(h)) != 0 and lib.a1mpc_sharded_handle(sh._h, -1, C.byref(h)) != 0     # shard out of range
        assert lib.a1mpc_sharded_handle(sh._h, 1, C.byref(h)) == 0 and h.value
        assert lib.a1mpc_sharded_last_transfer(None, None, None) != 0
        o = sh.solve_ticks(sc["tick"][:0], sc["R"][:0], sc["foot"][:0], sc["contact"][:0])                                      # n = 0: nothing to do
        assert o["grf"].shape == (0, 12)
    cfg1 = pkg.make_config(sc["params"], 1, warm_start=0)
    with pkg.ShardedEngine(cfg1, n, devices=[0], transport=0) as sh1:
        with pytest.raises(pkg.A1MpcError):
            sh1.solve_ticks(sc["tick"], sc["R"], sc["foot"], sc["contact"])                                                      # tick records need horizon >= 2
    with pkg.Pipeline(cfg, n, 0, depth=2) as pipe:
        ins = [t_(sc["x0"]), t_(sc["xref"]), t_(sc["R"]), t_(sc["foot"]), t_(sc["contact"], torch.uint8)]
        grf = torch.zeros(n, 12, dtype=torch.float64, device=dev)
        with pytest.raises(pkg.A1MpcError):
            pipe.submit_strided_device(n, ins[0], ins[1], ins[2], ins[3], 0, ins[4], 3, grf)                                    # contact_stride must be 0 or 4
        with pytest.raises(pkg.A1MpcError):
            pipe.submit_ticks_device(n, None, ins[2], ins[3], ins[4], grf)                                                      # null tick records
        with pytest.raises(pkg.A1MpcError):
            pipe.submit_strided_device(n + 1, *ins[:4], 0, ins[4], 0, grf)                                                       # n > max_batch
        k = pipe.submit_strided_device(n, *ins[:4], 0, ins[4], 0, grf, slot=1)                                                   # a fixed slot; (0, 0, NULL): the plain entry
        assert k == 1
        pipe.wait()
        assert float(grf.abs().max()) > 1.0
