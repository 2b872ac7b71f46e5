"""GPU: VERDICT r4 item 2 -- split pipeline (set-up kernel -> order -> persistent ADMM rows) against the fused kernel as persistent wavefronts on the same queue
(A1MPC_FUSED_QUEUE=1: a1mpc_solve_queue_kernel), 4096 x h10 cold first solves: one handle on one stream in three queue orders, then bench.py's pipelined `value`.
    python tools/fused_queue_probe.py [out.txt]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import bench
    pkg = bench.graft.load_package(); pkg.load_library()
    n = 4096; dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev)
    scs = [pkg.scenarios.config3_random_flat(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(4)]
    cfg = pkg.make_config(scs[0]["params"], 10, warm_start=0)
    ds = [{k: torch.from_numpy(s_[k]).to(dev) for k in ("x0", "xref", "R", "foot", "contact")} for s_ in scs]
    grf = torch.zeros((n, 12), dtype=torch.float64, device=dev); it = torch.zeros(n, dtype=torch.int32, device=dev); stt = torch.zeros(n, dtype=torch.int32, device=dev)
    out = {}
    with pkg.Engine(cfg, n, 0) as eng:
        def run(k, fresh):
            d = ds[k % 4]
            if fresh: eng.set_schedule(True)
            eng.solve_device(n, d["x0"], d["xref"], d["R"], d["foot"], d["contact"], grf, None, it, stt, stream=st.cuda_stream)
            return eng.last_kernel_ms()
        for k in range(8): run(k, True)
        out["first_solve_ms"] = float(np.median([run(k, True) for k in range(16)]))
        eng.set_schedule(False); run(0, False)
        out["index_order_ms"] = float(np.median([run(k, False) for k in range(16)]))
        eng.set_schedule(True); run(0, False)
        out["history_order_ms"] = float(np.median([run(0, False) for _ in range(16)]))
        run(0, True); torch.cuda.synchronize()
        out["digest_batch0"] = hashlib.sha256(grf.cpu().numpy().tobytes() + it.cpu().numpy().tobytes() + stt.cpu().numpy().tobytes()).hexdigest()[:16]
        out["mean_iters"] = float(it.float().mean().item())
    print("RESULT " + json.dumps(out))
    sys.exit(0)
lines = []
for flag in ("0", "1", "0", "1"):
    env = dict(os.environ, A1MPC_FUSED_QUEUE=flag)
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, timeout=300, env=env)
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    lines.append(f"A1MPC_FUSED_QUEUE={flag} one stream: " + (res[0][7:] if res else "FAILED " + r.stderr[-300:]))
    print(lines[-1], flush=True)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-latency", "--no-cpu-baseline", "--no-index-order"], capture_output=True, text=True,
                       timeout=300, env=dict(env, A1_BENCH_VALUE8="0"))
    try:
        d = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
        lines.append(f"A1MPC_FUSED_QUEUE={flag} bench.py (two batches in flight): value {d['value']:.0f} solves/s, ms_per_step {d['ms_per_step']:.4f}, single stream {d['roofline']['single_stream']['avg_kernel_ms']:.4f} ms")
    except Exception as e:
        lines.append(f"A1MPC_FUSED_QUEUE={flag} bench.py FAILED: {e} {b.stderr[-300:]}")
    print(lines[-1], flush=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
