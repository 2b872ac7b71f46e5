"""CPU, world_size 2, gloo: the multi-GPU path (contiguous partition, scatter inputs, gather GRFs).  The per-rank solver is
a stand-in here (there is no GPU): the ORACLE plays the engine -- test infrastructure standing in for the device, which is
exactly what may use oracle/ -- so the test checks the plumbing: every problem reaches exactly one rank and comes back in
order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    from helpers import oracle_params
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package(); O = g.load_oracle()
    sc = pkg.scenarios.config3_random_flat(nb=n)
    pr = oracle_params(O, sc)
    seen = []

    def solve_fn(loc):
        seen.append(len(loc["x0"]))
        r = O.mpc_solve_batch(pr, O.default_settings(), loc["x0"].numpy(), loc["xref"].numpy(), loc["R"].numpy(), loc["foot"].numpy(),
                              loc["contact"].numpy(), nthreads=1)
        return dict(grf=torch.from_numpy(r["grf"]), iters=torch.from_numpy(r["iters"]), status=torch.from_numpy(r["status"]))

    out = pkg.sharding.scatter_solve_gather(sc if rank == 0 else None, 10, solve_fn)
    if rank == 0:
        ref = O.mpc_solve_batch(pr, O.default_settings(), sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], nthreads=1)
        q.put((float(np.abs(out["grf"] - ref["grf"]).max()), bool((out["iters"] == ref["iters"]).all()), seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 2, 1])
def test_scatter_solve_gather_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, same, seen = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err == 0.0 and same
    assert seen == ([(n + 1) // 2] if n > 0 else [])


def test_partition(pkg):
    P = pkg.sharding.partition
    assert P(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert P(3, 8)[:4] == [(0, 1), (1, 1), (2, 1), (3, 0)]
    assert sum(c for _, c in P(65536, 8)) == 65536 and all(c == 8192 for _, c in P(65536, 8))


def _gpu_worker(rank, world, port, n, q):
    """two ranks share GPU 0 (the test box has one): communication over gloo host tensors, the per-rank solver is the ENGINE"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package()
    sc = pkg.scenarios.config3_random_flat(nb=n)
    cfg = pkg.make_config(sc["params"], 10, warm_start=0)
    cnt = pkg.sharding.partition(n, world)[rank][1]
    with pkg.Engine(cfg, max(cnt, 1), 0) as eng:
        def solve_fn(loc):
            r = eng.solve(loc["x0"].numpy(), loc["xref"].numpy(), loc["R"].numpy(), loc["foot"].numpy(), loc["contact"].numpy())
            return dict(grf=torch.from_numpy(r["grf"]), iters=torch.from_numpy(r["iters"]), status=torch.from_numpy(r["status"]))
        out = pkg.sharding.scatter_solve_gather(sc if rank == 0 else None, 10, solve_fn)
    if rank == 0:
        with pkg.Engine(cfg, n, 0) as eng:
            ref = eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"])
        q.put((bool(np.array_equal(out["grf"], ref["grf"])), bool(np.array_equal(out["iters"], ref["iters"])), bool((out["status"] == 1).all())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_and_run_the_engine():
    """the N > 1 path with the real engine as the per-rank solver (2 ranks on the one GPU of the test box)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, 601, q)) for r in range(2)]
    for p in procs:
        p.start()
    same_grf, same_it, solved = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same_grf and same_it and solved
