#!/usr/bin/env python3
"""A/B of two builds of liba1mpc.so on the same box: kernel ms at a few batch sizes (history order = the steadiest measurement).
usage: ab_probe.py libA.so libB.so [libC.so ...] [--only n] [--fixed k]"""
import ctypes as C, os, sys, subprocess, json
import numpy as np
if "--child" not in sys.argv:
    res = {}
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for lib in libs:
        out = subprocess.run([sys.executable, __file__, lib, "--child"] + rest, capture_output=True, text=True, timeout=90)
        res[os.path.basename(lib)] = json.loads(out.stdout.strip().splitlines()[-1])
    print(json.dumps(res, indent=1))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
pkg.engine._lib = None
pkg.engine._lib = pkg.engine.load_library(sys.argv[1]) if "r01" not in sys.argv[1] else None  # (load_library only caches the in-tree path)
if "r01" in sys.argv[1]:   # the round-1 library exports fewer symbols: bind by hand what the probe needs
    lib = C.CDLL(sys.argv[1]); pkg.engine._lib = None
    real = pkg.engine.load_library
    import types
    E = pkg.engine
    vp, i32 = C.c_void_p, C.c_int32
    dp = C.POINTER(C.c_double); u8p = C.POINTER(C.c_uint8); i32p = C.POINTER(C.c_int32)
    lib.a1mpc_default_config.argtypes = [C.POINTER(E.Config)]; lib.a1mpc_create.argtypes = [C.POINTER(E.Config), i32, i32, C.POINTER(vp)]
    lib.a1mpc_destroy.argtypes = [vp]; lib.a1mpc_solve_batch.argtypes = [vp, i32, dp, dp, dp, dp, u8p, dp, dp, i32p, i32p]
    lib.a1mpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]; lib.a1mpc_set_schedule.argtypes = [vp, i32]
    lib.a1mpc_status_string.restype = C.c_char_p; lib.a1mpc_last_error.restype = C.c_char_p
    E._lib = lib
    E.load_library = lambda path=None: lib
out = {}
sizes = (int(sys.argv[sys.argv.index('--only') + 1]),) if '--only' in sys.argv else (4096, 16384, 65536)
for n in sizes:
    sc = pkg.scenarios.config3_random_flat(nb=n)
    osqp = dict(warm_start=0)
    if '--fixed' in sys.argv:
        osqp.update(eps_abs=1e-300, eps_rel=1e-300, max_iter=int(sys.argv[sys.argv.index('--fixed') + 1]), adaptive_rho=0)
    cfg = pkg.make_config(sc["params"], 10, **osqp)
    with pkg.Engine(cfg, n, 0) as eng:
        ms = []
        for _ in range(6):
            eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); ms.append(eng.last_kernel_ms())
        eng.set_schedule(False); idx = []
        for _ in range(3):
            eng.solve(sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"]); idx.append(eng.last_kernel_ms())
    out[str(n)] = dict(history_ms=float(np.median(ms[2:])), index_ms=float(np.median(idx[1:])))
print(json.dumps(out))
