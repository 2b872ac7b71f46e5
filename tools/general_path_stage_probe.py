"""set-up | ADMM stage split of the general path (per-step feet + contact schedules) beyond its resident rows: a1mpc_last_stage_ms after a1mpc_solve_batch_strided"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.scenarios
for h, n in ((10, 4096), (10, 16384), (16, 8192), (20, 8192)):
    sc = S.config3_random_flat(nb=n, horizon=h); rng = np.random.default_rng(h)
    vd = rng.uniform(-0.6, 0.6, (n, 1, 1, 3))
    foot = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * sc["params"]["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12))
    sw = rng.integers(0, h + 1, (n, 4)); first = rng.integers(0, 2, (n, 4))
    contact = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :], 1 - first[:, None, :]).astype(np.uint8).reshape(n, h * 4))
    if os.environ.get("A1_PROBE_NO_RUIZ"):   # what of the set-up kernel is NOT the Ruiz passes (formation, gradient, hot state, hand-off record): the same launch with scaling = 0
        with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=0, scaling=0, max_iter=25), n, 0) as eng:
            st = []
            for _ in range(4):
                eng.set_schedule(True); eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4); st.append(eng.last_stage_ms())
            print(h, n, "   scaling = 0 (no Ruiz passes): set-up %.3f ms" % np.median(np.array(st[1:]), axis=0)[0])
    with pkg.Engine(pkg.make_config(sc["params"], h, warm_start=0), n, 0) as eng:
        st = []
        for _ in range(4):
            eng.set_schedule(True); eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4); st.append(eng.last_stage_ms())
        print(h, n, "general path first solve: set-up %.3f ms, ADMM %.3f ms" % tuple(np.median(np.array(st[1:]), axis=0)))
        st = []
        for _ in range(3):
            eng.solve_strided(sc["x0"], sc["xref"], sc["R"], foot, 12, contact, 4); st.append(eng.last_stage_ms())
        print(h, n, "              history order: set-up %.3f ms, ADMM %.3f ms" % tuple(np.median(np.array(st[1:]), axis=0)))
