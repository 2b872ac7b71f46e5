"""(study, CPU only) iteration counts of the general path's probe batch from the ORACLE -- data for general_path_cost_fit.py (profiles/r05_general_path_order.txt).
A tool, not product code: the oracle is test infrastructure."""
import os, sys, numpy as np, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.scenarios
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle as O
h = int(sys.argv[1]) if len(sys.argv) > 1 else 10; n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sc = S.config3_random_flat(nb=n, horizon=h)
rng = np.random.default_rng(h)
vd = rng.uniform(-0.6, 0.6, (n, 1, 1, 3))
foot = np.ascontiguousarray((sc["foot"].reshape(n, 1, 4, 3) - vd * sc["params"]["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(n, h * 12))
sw = rng.integers(0, h + 1, (n, 4)); first = rng.integers(0, 2, (n, 4))
contact = np.ascontiguousarray(np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :], 1 - first[:, None, :]).astype(np.uint8).reshape(n, h * 4))
P = sc["params"]
pr = O.mpc_params(h, P["dt"], P["mu"], P["fz_min"], P["fz_max"], P["q"], P["r"], P["mass"], P["inertia"])
st = O.default_settings()
def one(b):
    o = O.mpc_solve(pr, st, sc["x0"][b], sc["xref"][b], sc["R"][b], foot[b], contact[b], foot_stride=12, contact_stride=4)
    return o["info"].iters, o["info"].nfact if hasattr(o["info"], "nfact") else 0
if __name__ == "__main__":
    with mp.Pool(16) as pool: r = pool.map(one, range(n), chunksize=16)
    it = np.array([x[0] for x in r]); nf = np.array([x[1] for x in r])
    np.savez(f"gpurun_out/study_gen_h{h}.npz", it=it, nf=nf, x0=sc["x0"], xref=sc["xref"], contact=contact, foot=foot)
    print(h, n, it.mean(), it.min(), it.max(), nf.mean())
