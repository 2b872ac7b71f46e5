"""PCIe-inclusive rates of the host-pointer entries beside the device-resident rate: a1mpc_solve_batch (synchronous: snapshot into pinned memory, one copy in,
launches, one copy out) and a1mpc_pipeline_submit / _wait (the same per slot, but the caller's thread snapshots batch k + 1 while the GPU solves batch k).
First solves of distinct batches, like bench.py's `value`.  Checks that the pipelined outputs equal the synchronous call's bit for bit.
Usage: python tools/pcie_probe.py [n [horizon [steps [depths, e.g. 2,3]]]]   -> one JSON line"""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
gc.collect(); gc.disable()   # the caller of the reference is C++: a Python gen-2 collection (one 37 ms pause in a 25 ms timed loop, seen here) is not the library's
if os.environ.get("A1_LIB"):   # another build of the library (A/B)
    pkg.engine._lib = pkg.engine.load_library(os.environ["A1_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
gen = {10: pkg.scenarios.config3_random_flat, 16: pkg.scenarios.config4_random_h16, 20: pkg.scenarios.config5_divergent}[h]
scs = [gen(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(4)]
cfg = pkg.make_config(scs[0]["params"], h, warm_start=0)
res = dict(n=n, horizon=h, steps=steps, bytes_in_per_call=n * ((13 + 13 * h + 9 + 12) * 8 + 4), bytes_out_per_call=n * (12 * 8 + 8))
ref = []
with pkg.Engine(cfg, n, 0) as eng:
    for k in range(6):
        s = scs[k % 4]; eng.set_schedule(True); o = eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"])
        if k < 4: ref.append(o)
    t0 = time.perf_counter(); km = []
    for k in range(steps):
        s = scs[k % 4]
        eng.set_schedule(True); eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"]); km.append(eng.last_kernel_ms())
    ms = (time.perf_counter() - t0) / steps * 1e3
res["synchronous"] = dict(ms_per_call=ms, solves_per_s=n / ms * 1e3, kernel_ms=float(np.mean(km)))
for depth in ([int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else (2, 3)):
    pipe = pkg.engine.Pipeline(cfg, n, 0, depth=depth)
    outs = [dict(grf=np.zeros((n, 12)), iters=np.zeros(n, np.int32), status=np.zeros(n, np.int32)) for _ in range(depth)]
    same = True
    stamps = []
    def run(count, check=False):
        global same
        slots = [None] * depth
        for k in range(count):
            stamps.append(time.perf_counter())
            s = scs[k % 4]; j = k % depth
            if slots[j] is not None:
                pipe.wait(j)
                if check:
                    r = ref[slots[j]]
                    same = same and np.array_equal(outs[j]["grf"], r["grf"]) and np.array_equal(outs[j]["iters"], r["iters"]) and np.array_equal(outs[j]["status"], r["status"])
            pipe.submit(s["x0"], s["xref"], s["R"], s["foot"], s["contact"], outs[j], slot=j, fresh=True); slots[j] = k % 4
        pipe.wait(-1)
    run(2 * depth + 4, check=True)
    t0 = time.perf_counter(); run(steps); ms = (time.perf_counter() - t0) / steps * 1e3
    gaps = np.diff(np.array(stamps[-steps:])) * 1e3   # per-submit period of the timed loop (ms): a one-off stall shows here
    res[f"pipeline_depth{depth}"] = dict(ms_per_batch=ms, solves_per_s=n / ms * 1e3, bit_identical_to_synchronous=bool(same), median_period_ms=float(np.median(gaps)), max_period_ms=float(gaps.max()),
                                         periods_ms=[round(float(x), 2) for x in gaps] if os.environ.get("A1_PCIE_DEBUG") else None)
    pipe.close()
print(json.dumps(res))
