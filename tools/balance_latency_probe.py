import gc, os, sys, time, importlib
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
pkg = g.load_package(); scen = pkg.scenarios
gc.collect(); gc.disable()
sc = scen.balance_random(2000)
cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
with pkg.Engine(cfg, 8, 0) as eng:
    lat = np.zeros(2000)
    for t in range(2000):
        a = time.perf_counter(); eng.balance_solve(sc["root_acc"][t], sc["R"][t], sc["Rz"][t], sc["foot"][t], sc["contact"][t]); lat[t] = time.perf_counter() - a
l = lat[100:] * 1e3
print(f"balance QP, batch 1, ctypes, zero_copy_max={os.environ.get('A1MPC_ZERO_COPY_MAX','8')}: p50 {np.percentile(l,50):.4f} ms p99 {np.percentile(l,99):.4f} ms")
