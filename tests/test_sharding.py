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


def _participation_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    part = bench.participation(dist, rank, world, 0, "gloo")     # (a collective: every rank calls it; no GPU here -- the identity falls back to host + local ordinal)
    if rank == 0:
        q.put(part)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_participation_gather_world2():
    """round 6: the N > 1 bench line says which devices took part.  Two gloo ranks on this (GPU-less) host both claim local device 0: one distinct device, not a scaling point,
    both ranks listed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_participation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    part = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert part["world_size"] == 2 and part["distinct_devices"] == 1 and part["is_scaling_point"] is False and part["rccl_world_size"] == 0 and part["backend"] == "gloo"
    assert part["device_ids"].count("r0@") == 1 and part["device_ids"].count("r1@") == 1
    import bench
    one = bench.participation(None, 0, 1, 0, "nccl")              # N = 1: trivially a scaling point, no collective
    assert one["distinct_devices"] == 1 and one["is_scaling_point"] is True and one["rccl_world_size"] == 0


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


def _run_bench(*flags, timeout=420):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):   # no launcher around it: bench.py starts its own ranks
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself and prints n_gpus = 2 -- on the one-GPU test box the two ranks
    share the GPU over gloo (said so in config.parallelism); both the weak-scaling metric and --config 4 (scatter + solve + gather in the timed region) run end to end."""
    out = _run_bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-latency", "--no-index-order")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["solved_frac"] == 1.0
    assert "roofline" in out and out["unit"] == "solves/s"
    # round 6: the line proves which devices took part -- two ranks on ONE physical GPU are labelled, never counted as a scaling point
    c = out["config"]
    assert c["world_size"] == 2 and c["distinct_devices"] == 1 and c["is_scaling_point"] is False and c["rccl_world_size"] == 0 and c["backend"] == "gloo"
    assert c["device_ids"].count("r0@") == 1 and c["device_ids"].count("r1@") == 1 and "NOT a scaling point" in c["parallelism"]
    out4 = _run_bench("--gpus", "2", "--config", "4", "--batch", "4000", "--steps", "2", "--warmup", "1")
    assert out4["n_gpus"] == 2 and out4["scaling"] == "strong" and out4["config"]["global_batch"] == 4000 and out4["config"]["solved_frac"] > 0.99
    c4 = out4["config"]
    assert c4["distinct_devices"] == 1 and c4["is_scaling_point"] is False and c4["scatter_bytes_per_step"] == 2000 * ((13 + 13 * 16 + 9 + 12) * 8 + 4) and c4["gather_bytes_per_step"] == 2000 * 104
    assert c4["scatter_gather_ms_per_step_rank0"] > 0 and 0 < c4["scatter_gather_share_of_step"] < 1
    if os.environ.get("A1_KEEP_BENCH_LINES"):   # (profiles/r06_bench_2ranks_shared_gpu.json is this test's two lines)
        import json
        json.dump({"weak_scaling_line": out, "config4_line": out4}, open(os.environ["A1_KEEP_BENCH_LINES"], "w"), indent=1)
    one = _run_bench("--config", "4", "--batch", "4000", "--steps", "2", "--warmup", "1")   # N = 1: the device-resident path without host synchronisation in the step
    assert one["n_gpus"] == 1 and one["config"]["mean_iters"] == out4["config"]["mean_iters"]
    assert one["config"]["distinct_devices"] == 1 and one["config"]["is_scaling_point"] is True and one["config"]["pci_bus_id"] if "pci_bus_id" in one["config"] else True


@pytest.mark.gpu
def test_bench_native_runs_both_transports():
    """`bench.py --native t --gpus N`: one process, the batch sharded inside the C ABI (a1mpc_sharded_*), both transports measured side by side.  On the one-GPU box
    transport 0 runs its shards on the same GPU and the RCCL transport runs over the one device there is (communicator, root staging, no peer)."""
    out = _run_bench("--native", "0", "--gpus", "2", "--batch", "1000", "--steps", "3", "--warmup", "2")
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2000 and out["value"] > 0
    t0, t1 = out["transports"]["0"], out["transports"]["1"]
    assert t0["devices"] == [0, 0] and t0["solved_frac"] == 1.0
    assert "error" in t1 or (t1["solved_frac"] == 1.0 and t1["mean_iters"] == t0["mean_iters"]), t1


@pytest.mark.gpu
def test_eight_ranks_first_contact_on_the_shared_gpu():
    """VERDICT r3 item 8 (insurance for the first 8-GPU run; no scaling claim): the 8-rank case has never been started on real hardware, so everything about it that
    does not need eight devices is run here -- eight ranks x (engine handle + two-slot pipeline) on the one test GPU over gloo, self-spawned like the driver's command
    minus the launcher, for the weak-scaling metric and for --config 4 (scatter + solve + gather in the timed region), and eight shards through the native handle.
    Asserted: n_gpus = 8, every QP solved, the scatter / gather byte counts of the contiguous partition, and a wall-clock ceiling per call (rendezvous, eight torch
    imports and eight engine creations included)."""
    import time
    t0 = time.time()
    out = _run_bench("--gpus", "8", "--batch", "1024", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-latency", "--no-index-order", timeout=600)
    t_weak = time.time() - t0
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["solved_frac"] == 1.0 and out["config"]["batch_per_gpu"] == 1024
    N, H = 8192, 16
    t0 = time.time()
    out4 = _run_bench("--gpus", "8", "--config", "4", "--batch", str(N), "--steps", "2", "--warmup", "1", timeout=600)
    t_strong = time.time() - t0
    assert out4["n_gpus"] == 8 and out4["scaling"] == "strong" and out4["config"]["global_batch"] == N and out4["config"]["solved_frac"] == 1.0
    rec_bytes = (13 + 13 * H + 9 + 12) * 8 + 4     # x0, x_ref, R, feet as doubles + four contact bytes per QP (1 940 B at h = 16)
    away = N - N // 8                              # QPs that leave rank 0
    assert out4["scatter_bytes_per_step"] == away * rec_bytes and out4["gather_bytes_per_step"] == away * (12 * 8 + 8), (out4["scatter_bytes_per_step"], out4["gather_bytes_per_step"])
    t0 = time.time()
    nat = _run_bench("--native", "0", "--gpus", "8", "--batch", "512", "--steps", "3", "--warmup", "2", timeout=600)
    t_nat = time.time() - t0
    assert nat["n_gpus"] == 8 and nat["config"]["global_batch"] == 8 * 512 and nat["transports"]["0"]["devices"] == [0] * 8 and nat["transports"]["0"]["solved_frac"] == 1.0
    print(f"8 ranks on one GPU: weak {t_weak:.0f} s, config 4 {t_strong:.0f} s, native 8 shards {t_nat:.0f} s wall clock")
    assert max(t_weak, t_strong) < 420 and t_nat < 180, (t_weak, t_strong, t_nat)
