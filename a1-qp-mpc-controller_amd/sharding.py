"""Batch sharding across the GPUs of one node.

The QPs of a batch are independent, so the partition is contiguous (remainder to the low ranks) and there is no
data-path collective.  When the whole batch lives on rank 0, inputs are scattered and GRFs gathered with
`torch.distributed` point-to-point-backed collectives -- backend "nccl" (RCCL over xGMI) on the GPUs, "gloo" in the CPU
test.  `solve_fn(local_inputs) -> dict(grf, iters, status)` is the per-rank solver (an Engine bound to that rank's GPU).
"""
import numpy as np
import torch
import torch.distributed as dist

FIELDS = (("x0", 13, torch.float64), ("xref", None, torch.float64), ("R", 9, torch.float64), ("foot", 12, torch.float64),
          ("contact", 4, torch.uint8))


def partition(n, world):
    """[(start, count)] per rank: contiguous, sizes differ by at most one, remainder to the low ranks."""
    base, rem = divmod(n, world)
    out, s = [], 0
    for r in range(world):
        c = base + (1 if r < rem else 0)
        out.append((s, c)); s += c
    return out


def scatter_solve_gather(inputs, horizon, solve_fn, device="cpu", group=None, root=0):
    """inputs: dict of numpy arrays on `root` (ignored elsewhere).  Returns dict(grf, iters, status) on root, None elsewhere."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = [len(inputs["x0"]) if rank == root else 0]
    dist.broadcast_object_list(meta, src=root, group=group)
    n = meta[0]
    parts = partition(n, world)
    start, cnt = parts[rank]
    local = {}
    for name, width, dtype in FIELDS:
        w = 13 * horizon if width is None else width
        buf = torch.empty((cnt, w), dtype=dtype, device=device)
        if rank == root:
            full = torch.from_numpy(np.ascontiguousarray(inputs[name]).reshape(n, w)).to(device)
            chunks = [full[s:s + c].contiguous() for s, c in parts]
            buf.copy_(chunks[root])
            reqs = [dist.isend(chunks[r], dst=r, group=group) for r in range(world) if r != root and parts[r][1] > 0]
            for q in reqs:
                q.wait()
        elif cnt > 0:
            dist.recv(buf, src=root, group=group)
        local[name] = buf
    res = solve_fn(local) if cnt > 0 else dict(grf=torch.empty((0, 12), dtype=torch.float64, device=device),
                                                iters=torch.empty(0, dtype=torch.int32, device=device),
                                                status=torch.empty(0, dtype=torch.int32, device=device))
    out = None
    if rank == root:
        out = dict(grf=torch.empty((n, 12), dtype=torch.float64, device=device), iters=torch.empty(n, dtype=torch.int32, device=device),
                   status=torch.empty(n, dtype=torch.int32, device=device))
    for key in ("grf", "iters", "status"):
        t = torch.as_tensor(res[key], device=device).contiguous()
        if rank == root:
            out[key][start:start + cnt].copy_(t)
            for r in range(world):
                if r != root and parts[r][1] > 0:
                    dist.recv(out[key][parts[r][0]:parts[r][0] + parts[r][1]], src=r, group=group)
        elif cnt > 0:
            dist.send(t, dst=root, group=group)
    if rank == root:
        return {k: v.cpu().numpy() for k, v in out.items()}
    return None
