#!/usr/bin/env python3
"""Timeline model of the persistent ADMM kernel: 1024 waves x ROWS rows drain the queue; the rows of a wave share one instruction
stream, so a factor pass or a QP hand-over of one row stalls its wave-mates unless they do the same thing at the same boundary.
Input: gpurun_out/iters_dump.npz (tools/dump_iters.py).  Compares: measured, model, model with independent rows (no sharing stalls).
Assumption: the factor passes of a QP happen at its first checkpoints (rho settles early)."""
import sys, heapq
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/iters_dump.npz")
T_SEG, T_FAC, T_IO = 25 * 3.1 + 4.0, 28.0, 14.0   # us: 25 iterations + check, one factor pass, hand-over (write + queue + load)
WAVES = 1024


def run(iters, nfact, order, rows, shared=True):
    segs = (iters // 25).astype(int)
    q = list(order)
    qi = 0
    waves = [(0.0, w) for w in range(WAVES)]
    state = {w: [None] * rows for w in range(WAVES)}  # per row: [segments left, factor passes left] or None
    heapq.heapify(waves)
    end = 0.0
    while waves:
        t, w = heapq.heappop(waves)
        st = state[w]
        cost_io = cost_f = 0.0
        n_io = n_f = 0
        for r in range(rows):
            if st[r] is None or st[r][0] == 0:
                had = st[r] is not None
                if qi < len(q):
                    j = q[qi]; qi += 1
                    st[r] = [segs[j], nfact[j]]
                    n_io += 1
                else:
                    st[r] = None
                    n_io += 1 if had else 0
        if all(s is None for s in st):
            end = max(end, t + (T_IO if n_io else 0.0))
            continue
        for r in range(rows):
            if st[r] is not None and st[r][1] > 0:
                st[r][1] -= 1; n_f += 1
        if shared:
            dt = (T_IO if n_io else 0.0) + (T_FAC if n_f else 0.0) + T_SEG
        else:  # independent rows: a wave-step costs the average of what its rows need (no stall of the mates)
            live = sum(s is not None for s in st)
            dt = (T_IO * n_io + T_FAC * n_f) / max(live, 1) + T_SEG
        for r in range(rows):
            if st[r] is not None:
                st[r][0] -= 1
        heapq.heappush(waves, (t + dt, w))
    return end / 1000.0


for n in (4096, 16384):
    it, nf = d[f"iters_{n}"], d[f"nfact_{n}"]
    cost = it + 10 * nf
    hist = np.argsort(-cost, kind="stable")
    print(f"n={n}: measured K1+K2 history {float(d[f'ms_{n}']):.3f} ms, index {float(d[f'ms_index_{n}']):.3f} ms (K1 ~ {0.153 if n == 4096 else 0.43})")
    for name, order in (("history", hist), ("index", np.arange(n))):
        print(f"   {name:8s} model K2: rows share a stream {run(it, nf, order, 2):.3f} ms | independent rows {run(it, nf, order, 2, shared=False):.3f} ms")
    print(f"   mean iters {it.mean():.1f}, mean factor passes {nf.mean():.2f}, max iters {it.max()}")
