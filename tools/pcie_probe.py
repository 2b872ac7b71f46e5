"""PCIe-inclusive rate of the host-pointer entry (a1mpc_solve_batch: snapshot into pinned memory, one copy in, launch, one copy out) beside the device-resident rate.
Usage: python tools/pcie_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scs = [pkg.scenarios.config3_random_flat(nb=n, seed=0xA1 + 3 + 17 * k) for k in range(4)]
cfg = pkg.make_config(scs[0]["params"], 10, warm_start=0)
with pkg.Engine(cfg, n, 0) as eng:
    for k in range(6):
        eng.set_schedule(True); eng.solve(scs[k % 4]["x0"], scs[k % 4]["xref"], scs[k % 4]["R"], scs[k % 4]["foot"], scs[k % 4]["contact"])
    t0 = time.perf_counter(); km = []
    for k in range(40):
        s = scs[k % 4]
        eng.set_schedule(True); eng.solve(s["x0"], s["xref"], s["R"], s["foot"], s["contact"]); km.append(eng.last_kernel_ms())
    ms = (time.perf_counter() - t0) / 40 * 1e3
print(f"{n} x h10 first solves through host pointers: {ms:.3f} ms per call = {n / ms / 1e3:.2f} M solves/s (kernels {np.mean(km):.3f} ms of it; "
      f"{n * (13 + 130 + 9 + 12) * 8 + n * 4} B in, {n * (12 * 8 + 8)} B out per call)")
