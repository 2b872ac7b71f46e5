"""GPU: up to which batch size does the latency kernel (one QP per wavefront, its four rows share the set-up) beat the fused kernel (two QPs per wavefront) at h = 10?
Children of tools/env_ab.py under A1MPC_COOP_MAX = 256 (latency kernel up to 256 QPs) and 2048, cold first solves and warm-started ticks; digests compared.
-> profiles/r05_latency_kernel_batch_limit.txt"""
import json, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
for warm, ticks in ((0, 4), (1, 12), (2, 12)):
    for n in (256, 384, 512, 768, 1024, 1536, 2048):
        row = {}
        for lim in ("256", "2048"):
            r = subprocess.run([sys.executable, os.path.join(HERE, "env_ab.py"), "child", str(n), str(ticks), str(warm)], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, A1MPC_COOP_MAX=lim))
            res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            row[lim] = json.loads(res[0][7:]) if res else {"error": r.stderr[-300:]}
        a, b = row["256"], row["2048"]
        if "error" in a or "error" in b:
            print(n, warm, a.get("error"), b.get("error")); continue
        ka, kb = np.array(a["kernel_ms"]), np.array(b["kernel_ms"])
        sl = slice(1, None) if warm == 0 else slice(3, None)
        print(f"warm_start {warm}  n {n:5d}  limit 256: {np.median(ka[sl]):.4f} ms   limit 2048: {np.median(kb[sl]):.4f} ms   same bits: {a['digest'] == b['digest']}  mean iters {a['mean_iters']:.1f}", flush=True)
