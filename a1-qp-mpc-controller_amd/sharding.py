"""Batch sharding across the GPUs of one node, one process per GPU.

The QPs of a batch are independent, so the partition is contiguous (remainder to the low ranks) and there is no
data-path collective.  When the whole batch lives on rank 0, inputs are scattered and results gathered with ONE group
of point-to-point operations each way (`torch.distributed.batch_isend_irecv` = grouped ncclSend / ncclRecv on backend
"nccl", i.e. RCCL over xGMI: the root drives all its links concurrently; "gloo" in the CPU test): per peer two sends out
(the float64 record [x0 | x_ref | R | foot] and the contact bytes) and two back ([grf] and [iters, status]).
`solve_fn(local_inputs) -> dict(grf, iters, status)` is the per-rank solver (an Engine bound to that rank's GPU).
The single-process equivalent inside the C ABI is a1mpc_sharded_* (include/a1mpc.h).
"""
import numpy as np
import torch
import torch.distributed as dist

FIELDS = (("x0", 13), ("xref", None), ("R", 9), ("foot", 12))


def partition(n, world):
    """[(start, count)] per rank: contiguous, sizes differ by at most one, remainder to the low ranks."""
    base, rem = divmod(n, world)
    out, s = [], 0
    for r in range(world):
        c = base + (1 if r < rem else 0)
        out.append((s, c)); s += c
    return out


def record_width(horizon):
    return 13 + 13 * horizon + 9 + 12


def pack_inputs(inputs, horizon, device="cpu"):
    """the batch as it lives on the root: one float64 record per QP + the contact bytes (device-resident tensors)"""
    n = len(inputs["x0"])
    rec = torch.cat([torch.as_tensor(np.ascontiguousarray(inputs[k]).reshape(n, 13 * horizon if w is None else w), dtype=torch.float64) for k, w in FIELDS], dim=1)
    ct = torch.as_tensor(np.ascontiguousarray(inputs["contact"]).reshape(n, 4), dtype=torch.uint8)
    return rec.contiguous().to(device), ct.contiguous().to(device)


def unpack_record(rec, horizon):
    o, out = 0, {}
    for k, w in FIELDS:
        w = 13 * horizon if w is None else w
        out[k] = rec[:, o:o + w].contiguous(); o += w
    return out


def _grouped(ops, group):
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()


def scatter(rec, ct, n, horizon, device="cpu", group=None, root=0):
    """root: (rec, ct) hold the whole batch; returns this rank's (rec, ct) slices (views on the root, fresh tensors elsewhere)"""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    parts = partition(n, world)
    start, cnt = parts[rank]
    if rank == root:
        ops = []
        for r in range(world):
            if r != root and parts[r][1] > 0:
                s, c = parts[r]
                ops += [dist.P2POp(dist.isend, rec[s:s + c], r, group), dist.P2POp(dist.isend, ct[s:s + c], r, group)]
        _grouped(ops, group)
        return rec[start:start + cnt], ct[start:start + cnt]
    lrec = torch.empty((cnt, record_width(horizon)), dtype=torch.float64, device=device)
    lct = torch.empty((cnt, 4), dtype=torch.uint8, device=device)
    if cnt > 0:
        _grouped([dist.P2POp(dist.irecv, lrec, root, group), dist.P2POp(dist.irecv, lct, root, group)], group)
    return lrec, lct


def gather(grf, meta, n, device="cpu", group=None, root=0, out=None):
    """grf (cnt, 12) float64 and meta (cnt, 2) int32 = [iters, status] of this rank -> the whole batch on root (None elsewhere)"""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    parts = partition(n, world)
    start, cnt = parts[rank]
    if rank == root:
        og, om = out if out is not None else (torch.empty((n, 12), dtype=torch.float64, device=device), torch.empty((n, 2), dtype=torch.int32, device=device))
        og[start:start + cnt].copy_(grf); om[start:start + cnt].copy_(meta)
        ops = []
        for r in range(world):
            if r != root and parts[r][1] > 0:
                s, c = parts[r]
                ops += [dist.P2POp(dist.irecv, og[s:s + c], r, group), dist.P2POp(dist.irecv, om[s:s + c], r, group)]
        _grouped(ops, group)
        return og, om
    if cnt > 0:
        _grouped([dist.P2POp(dist.isend, grf.contiguous(), root, group), dist.P2POp(dist.isend, meta.contiguous(), root, group)], group)
    return None


def scatter_solve_gather(inputs, horizon, solve_fn, device="cpu", group=None, root=0):
    """inputs: dict of numpy arrays on `root` (ignored elsewhere).  Returns dict(grf, iters, status) on root, None elsewhere."""
    rank = dist.get_rank(group)
    meta = [len(inputs["x0"]) if rank == root else 0]
    dist.broadcast_object_list(meta, src=root, group=group)
    n = meta[0]
    rec, ct = pack_inputs(inputs, horizon, device) if rank == root else (None, None)
    lrec, lct = scatter(rec, ct, n, horizon, device, group, root)
    cnt = lrec.shape[0]
    if cnt > 0:
        local = unpack_record(lrec, horizon); local["contact"] = lct
        res = solve_fn(local)
        grf = torch.as_tensor(res["grf"], device=device).reshape(cnt, 12).to(torch.float64)
        m = torch.stack([torch.as_tensor(res["iters"], device=device).to(torch.int32), torch.as_tensor(res["status"], device=device).to(torch.int32)], dim=1)
    else:
        grf = torch.empty((0, 12), dtype=torch.float64, device=device); m = torch.empty((0, 2), dtype=torch.int32, device=device)
    out = gather(grf, m, n, device, group, root)
    if rank == root:
        og, om = out
        return dict(grf=og.cpu().numpy(), iters=om[:, 0].cpu().numpy().copy(), status=om[:, 1].cpu().numpy().copy())
    return None
