# numpy study of the Ruiz sweep pruning bounds (run from tests/: python tools/prune_study.py); uses tests/ref_numpy.py
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.getcwd())); sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
import ref_numpy as rn
pkg = g.load_package()
H = 10
NQ = 8
sc = pkg.scenarios.config3_random_flat(nb=NQ)
al = np.zeros((H, H)); be = np.zeros((H, H))
for s in range(H):
    for t in range(H):
        m = max(s, t); al[s, t] = sum((i - s) * (i - t) for i in range(m, H)); be[s, t] = H - m
gam = al / be
def ruiz_D(P, A, passes=10):
    n, m = P.shape[0], A.shape[0]
    D = np.ones(n); E = np.ones(m)
    Ds = []
    Pc, Ac = P.copy(), A.copy()
    for _ in range(passes):
        Ds.append(D.copy())
        dn = np.maximum(np.abs(Pc).max(0), np.abs(Ac).max(0)); dn = np.where(dn < 1e-4, 1.0, np.minimum(dn, 1e4)); dt = 1 / np.sqrt(dn)
        en = np.abs(Ac).max(1); en = np.where(en < 1e-4, 1.0, np.minimum(en, 1e4)); et = 1 / np.sqrt(en)
        Pc = dt[:, None] * Pc * dt[None, :]; Ac = et[:, None] * Ac * dt[None, :]
        D *= dt; E *= et
        cn = min(max(np.abs(Pc).max(0).mean(), 1e-4), 1e4)
        Pc *= 1 / cn
    Ds.append(D.copy())
    return Ds
allQ = []
for q in range(NQ):
    P, gq, A, l, u = rn.mpc_qp(sc["params"], H, sc["x0"][q], sc["xref"][q], sc["R"][q].reshape(3, 3), sc["foot"][q].reshape(4, 3), sc["contact"][q])
    X1 = P[0:12, 12:24] / be[0, 1]; X2 = P[0:12, 24:36] / be[0, 2]
    U = (X1 - X2) / (gam[0, 1] - gam[0, 2]); V = X1 - gam[0, 1] * U
    assert np.allclose(P[36:48, 60:72], be[3, 5] * (gam[3, 5] * U + V), rtol=1e-8, atol=1e-8)
    allQ.append((np.abs(P), np.abs(U), np.abs(V), ruiz_D(P, A)))
tot = dict(blocks=0, needed=0, cur=0, tight=0, wblocks=0, need_w=0, cur_w=0, tight_w=0)
for wave in range(NQ // 4):
    qs = allQ[4 * wave: 4 * wave + 4]
    for p in range(11):
        need = np.zeros((4, H, 12, H), bool); cur = np.zeros_like(need); tight = np.zeros_like(need)
        for k, (Pabs, Ua, Va, Ds) in enumerate(qs):
            D = Ds[p].reshape(H, 12)
            Dmax = D.max(1)
            UD = (Ua[:, None, :] * D[None, :, :]).max(2)  # [a][t]
            VD = (Va[:, None, :] * D[None, :, :]).max(2)
            Umax = Ua.max(1); Vmax = Va.max(1)
            for s in range(H):
                for a in range(12):
                    i = s * 12 + a
                    mm = Pabs[i, i] * D[s, a]
                    Eall = (Pabs[i].reshape(H, 12) * D).max(1)
                    for t in range(H):
                        bA = (gam[s, t] * Umax[a] + Vmax[a]) * be[s, t] * Dmax[t]
                        bB = be[s, t] * (gam[s, t] * UD[a, t] + VD[a, t])
                        need[k, s, a, t] = Eall[t] > mm; cur[k, s, a, t] = bA > mm; tight[k, s, a, t] = bB > mm
                        mm = max(mm, Eall[t])
        tot["blocks"] += need.size; tot["needed"] += need.sum(); tot["cur"] += cur.sum(); tot["tight"] += tight.sum()
        tot["wblocks"] += H * H; tot["need_w"] += need.any(axis=(0, 2)).sum(); tot["cur_w"] += cur.any(axis=(0, 2)).sum(); tot["tight_w"] += tight.any(axis=(0, 2)).sum()
print("lane level: needed %.3f, current bound executes %.3f, tight bound %.3f" % (tot["needed"] / tot["blocks"], tot["cur"] / tot["blocks"], tot["tight"] / tot["blocks"]))
print("wave level (4 QPs x 12 lanes must agree): needed %.3f, current %.3f, tight %.3f" % (tot["need_w"] / tot["wblocks"], tot["cur_w"] / tot["wblocks"], tot["tight_w"] / tot["wblocks"]))
