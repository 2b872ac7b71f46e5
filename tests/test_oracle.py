"""CPU: pins the oracle (oracle/a1mpc_oracle.c).  The reference pins nothing (S/test/test_mpc.cpp:157-161 only prints) and
cannot be built here, so the oracle is pinned by (1) an independent numpy re-formation of the QP matrices after the
reference's own formulas, (2) the KKT conditions of its tight-mode solutions on those matrices, (3) an independent scipy
solve, (4) analytic stand cases, (5) the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

import ref_numpy as RN
from helpers import oracle_batch, oracle_params

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _one(sc, b=0):
    h = sc["horizon"]
    return sc["x0"][b], sc["xref"][b], sc["R"][b].reshape(3, 3), sc["foot"][b].reshape(4, 3), sc["contact"][b]


@pytest.mark.parametrize("gen,kw", [("scenario_T", {}), ("config3_random_flat", dict(nb=3)), ("config4_random_h16", dict(nb=2)),
                                     ("config5_divergent", dict(nb=2)), ("config3_random_flat", dict(nb=2, param_set="hardware"))])
def test_formation_matches_independent_numpy(oracle, scen, gen, kw):
    sc = getattr(scen, gen)(**kw)
    pr = oracle_params(oracle, sc)
    for b in range(len(sc["x0"])):
        x0, xref, R, foot, contact = _one(sc, b)
        P, g, A, l, u, _ = oracle.mpc_form(pr, x0, xref, sc["R"][b], sc["foot"][b], contact)
        P2, g2, A2, l2, u2 = RN.mpc_qp(sc["params"], sc["horizon"], x0, xref, R, foot, contact)
        assert np.abs(P - P2).max() <= 1e-12 * np.abs(P2).max()
        assert np.abs(g - g2).max() <= 1e-11 * max(1.0, np.abs(g2).max())
        assert (A == A2).all() and (l == l2).all() and (u == u2).all()


@pytest.mark.parametrize("name", ["T_test_mpc", "stand_gazebo", "config3_h10", "config4_h16", "config5_h20", "config3_hardware_weights"])
def test_exact_mode_solutions_satisfy_kkt(scen, name):
    """the `exact` golden solutions are the QP optimum: stationarity, feasibility, multiplier signs on numpy matrices"""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ps = "hardware" if "hardware" in name else ("test_mpc" if name.startswith("T_") else ("isaac" if "isaac" in name else "gazebo"))
    params = dict(scen.PARAM_SETS[ps], **scen.MPC_CONSTANTS)
    h = int(z["horizon"])
    for b in range(min(4, len(z["x0"]))):
        P, g, A, l, u = RN.mpc_qp(params, h, z["x0"][b], z["xref"][b], z["R"][b].reshape(3, 3), z["foot"][b].reshape(4, 3), z["contact"][b])
        stat, viol, wrong = RN.kkt_violation(P, g, A, l, u, z["exact_u"][b], tol_active=1e-5)
        scale = max(1.0, np.abs(g).max())
        assert stat <= 1e-6 * scale and viol <= 1e-6 and wrong <= 1e-6 * scale, (name, b, stat, viol, wrong)


def test_scipy_cross_check_fixture_T(oracle, scen):
    """independent solver (scipy SLSQP, a dense active-set SQP) on the numpy matrices of fixture T agrees with the oracle's exact mode"""
    from scipy.optimize import minimize
    sc = scen.scenario_T()
    x0, xref, R, foot, contact = _one(sc)
    P, g, A, l, u = RN.mpc_qp(sc["params"], 10, x0, xref, R, foot, contact)
    fin_l = l > -1e20; fin_u = u < 1e20
    Ai = np.vstack([A[fin_l], -A[fin_u]]); bi = np.concatenate([-l[fin_l], u[fin_u]])
    res = minimize(lambda x: 0.5 * x @ P @ x + g @ x, np.zeros(120), jac=lambda x: P @ x + g, method="SLSQP",
                   constraints=[dict(type="ineq", fun=lambda x: Ai @ x + bi, jac=lambda x: Ai)], options=dict(ftol=1e-14, maxiter=500))
    assert res.status == 0
    e = oracle_batch(oracle, sc, settings=oracle.exact_settings())
    assert np.abs(res.x[:12] - e["u"][0][:12]).max() < 5e-3  # SLSQP's own accuracy (observed 4e-4); survey estimate (0,-12.837,42.790)
    f = e["grf"][0].reshape(4, 3)
    assert f[0, 2] == pytest.approx(42.790, abs=2e-3) and f[0, 1] == pytest.approx(-12.837, abs=2e-3)


def test_analytic_stand_cases(oracle, scen):
    for ps in ("gazebo", "isaac"):
        sc = scen.scenario_stand(ps)
        e = oracle_batch(oracle, sc, settings=oracle.exact_settings())
        fz = e["grf"][0].reshape(4, 3)[:, 2]
        assert fz.sum() == pytest.approx(sc["params"]["mass"] * 9.8, rel=2e-3)
    s1 = scen.config1_balance_stand()
    r = oracle.balance_solve(oracle.default_qp_params(), oracle.default_settings(), s1["root_acc"][0], s1["R"][0], s1["Rz"][0], s1["foot"][0], s1["contact"][0])
    assert np.allclose(r["grf"].reshape(4, 3)[:, 2], 12 * 9.8 / 4, atol=0.02)


def test_balance_formation_and_kkt(oracle, scen):
    sc = scen.balance_random(6)
    qp = oracle.default_qp_params()
    for b in range(6):
        P, g, A, l, u, _ = oracle.balance_form(qp, sc["root_acc"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
        P2, g2, A2, l2, u2 = RN.balance_qp(sc["root_acc"][b], sc["Rz"][b].reshape(3, 3), sc["foot"][b].reshape(4, 3), sc["contact"][b])
        assert np.abs(P - P2).max() < 1e-10 and np.abs(g - g2).max() < 1e-8 and (A == A2).all() and (l == l2).all() and (u == u2).all()
        r = oracle.balance_solve(qp, oracle.exact_settings(), sc["root_acc"][b], sc["R"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
        stat, viol, wrong = RN.kkt_violation(P2, g2, A2, l2, u2, r["f_world"], tol_active=1e-6)
        assert stat < 1e-5 and viol < 1e-7 and wrong < 1e-5


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_golden(oracle, scen, path):
    z = np.load(path)
    name = os.path.basename(path)[:-4]
    if name == "balance_random":
        qp, st = oracle.default_qp_params(), oracle.default_settings()
        for b in range(len(z["root_acc"])):
            r = oracle.balance_solve(qp, st, z["root_acc"][b], z["R"][b], z["Rz"][b], z["foot"][b], z["contact"][b])
            assert r["info"].iters == z["default_iters"][b] and np.abs(r["f_world"] - z["default_f"][b]).max() < 1e-7
        return
    ps = "hardware" if "hardware" in name else ("test_mpc" if name.startswith("T_") else ("isaac" if "isaac" in name else "gazebo"))
    sc = dict(params=dict(scen.PARAM_SETS[ps], **scen.MPC_CONSTANTS), horizon=int(z["horizon"]), x0=z["x0"], xref=z["xref"], R=z["R"], foot=z["foot"],
              contact=z["contact"])
    if name == "config2_warm_sequence":
        pr = oracle_params(oracle, sc); st = oracle.default_settings(warm_start=1)
        wx = np.zeros(120); wy = np.zeros(200); rho = None
        for t in range(len(z["x0"])):
            r = oracle.mpc_solve(pr, st, z["x0"][t], z["xref"][t], z["R"][t], z["foot"][t], z["contact"][t], warm_x=wx, warm_y=wy, warm_rho=rho)
            wx, wy, rho = r["warm_x"], r["warm_y"], r["rho"]
            assert r["info"].iters == z["iters"][t] and np.abs(r["u"] - z["u"][t]).max() < 1e-8
        return
    d = oracle_batch(oracle, sc)
    assert (d["iters"] == z["default_iters"]).all() and np.abs(d["u"] - z["default_u"]).max() < 1e-8
    e = oracle_batch(oracle, sc, settings=oracle.exact_settings())
    assert np.abs(e["u"] - z["exact_u"]).max() < 1e-6


def test_update_plan_restatement(oracle):
    """N2a oracle vs a direct numpy transcription of S/A1RobotControl.cpp:148-202 on hand-picked cases"""
    dfp = np.array([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35])
    gp = oracle.gait_params(dfp)
    Rz = np.eye(3).reshape(9); R = np.eye(3).reshape(9)
    # stand: counters reset, all feet planned in contact, no velocity -> default footholds
    gc, pc, rel, ab, wo = oracle.update_plan(gp, 0, [5, 6, 7, 8], [2, 2, 2, 2], [0, 0, 0], Rz, R, [1, 2, 3], [0, 0, 0])
    assert list(gc) == [0, 120, 120, 0] and list(pc) == [1, 1, 1, 1] and (rel == dfp).all() and np.allclose(wo.reshape(4, 3), dfp.reshape(4, 3) + [1, 2, 3])
    # walk: counter advance, wrap and swing threshold
    gc, pc, rel, ab, wo = oracle.update_plan(gp, 1, [119, 120, 239, 0], [2, 2, 2, 2], [0.5, 0, 0], Rz, R, [0, 0, 0], [0.5, 0, 0])
    assert list(gc) == [121, 122, 1, 2] and list(pc) == [0, 0, 1, 1]
    dx = np.sqrt(0.35 / 9.8) * 0.0 + ((120 / 2.0) * 0.0025) / 2.0 * 0.5
    assert np.allclose(rel.reshape(4, 3)[:, 0], dfp.reshape(4, 3)[:, 0] + dx, rtol=0, atol=1e-16)
    # saturation of the foothold offsets
    gc, pc, rel, ab, wo = oracle.update_plan(gp, 1, [0, 0, 0, 0], [2, 2, 2, 2], [5.0, -5.0, 0], Rz, R, [0, 0, 0], [0, 0, 0])
    assert np.allclose(rel.reshape(4, 3)[:, 0] - dfp.reshape(4, 3)[:, 0], 0.1) and np.allclose(rel.reshape(4, 3)[:, 1] - dfp.reshape(4, 3)[:, 1], -0.1)


def test_joint_torques_restatement(oracle):
    """N3 oracle: stance = J'(-f), swing solves J tau = km .* f_kin (checked against numpy's LAPACK solve), inactive = zeros"""
    rng = np.random.default_rng(3)
    km = np.array([0.1, 0.1, 0.04])
    for trial in range(200):
        Jb = rng.normal(0, 0.3, (4, 3, 3))  # [leg][row][col]
        Jcm = np.stack([Jb[i].T.reshape(9) for i in range(4)]).reshape(36)  # column-major blocks
        c = (rng.random(4) < 0.5).astype(np.uint8); grf = rng.normal(0, 40, 12); fk = rng.normal(0, 20, 12); tg = rng.normal(0, 1, 12)
        tau = oracle.joint_torques(1, c, Jcm, grf, fk, km, tg, np.zeros(12))
        for i in range(4):
            ref = Jb[i].T @ -grf[3 * i:3 * i + 3] if c[i] else np.linalg.solve(Jb[i], km * fk[3 * i:3 * i + 3])
            assert np.allclose(tau[3 * i:3 * i + 3], ref + tg[3 * i:3 * i + 3], rtol=1e-9, atol=1e-9)
    assert (oracle.joint_torques(0, c, Jcm, grf, fk, km, tg, np.ones(12)) == 0).all()


def test_contact_terrain_restatement(oracle):
    """N2b oracle: moving-window filters vs a plain deque mean, plane fit vs numpy's SVD pinv, early-contact / terrain rules"""
    import collections
    rng = np.random.default_rng(5)
    st = oracle.contact_state()
    dq = [collections.deque(maxlen=60) for _ in range(12)]; tq = collections.deque(maxlen=100)
    rec_ref = np.zeros(12); early = np.zeros(4, bool); pitch = 0.0
    for tick in range(400):
        gc = rng.uniform(0, 240, 4); plan = (gc <= 120).astype(np.uint8); ff = rng.uniform(0, 80, 4)
        foot = rng.normal(0, 0.05, 12) + np.tile([0.0, 0.0, -0.3 + 0.05 * np.sin(tick / 40)], 4) + np.outer([0.2, 0.2, -0.2, -0.2], [1.0, 0.0, 0.3]).reshape(12) \
            + np.outer([1, -1, 1, -1], [0.0, 0.13, 0.0]).reshape(12)
        z = 0.3 if tick % 50 else 0.05
        ct, rec, ang, pitch = oracle.contact_terrain_step(st, gc, plan, ff, foot, z, pitch)
        for i in range(4):
            if gc[i] <= 180: early[i] = False
            if (not plan[i]) and gc[i] > 180 and ff[i] > 30: early[i] = True
            c = bool(plan[i]) or early[i]
            assert ct[i] == c
            if c:
                for k in range(3):
                    dq[3 * i + k].append(foot[3 * i + k]); rec_ref[3 * i + k] = sum(dq[3 * i + k]) / 60.0
        assert np.allclose(rec, rec_ref, rtol=0, atol=1e-13)
        W = np.stack([np.ones(4), rec_ref[0::3], rec_ref[1::3]], 1); a = np.linalg.pinv(W.T @ W) @ W.T @ rec_ref[2::3]
        if z > 0.1:
            tq.append(np.arccos(1.0 / np.sqrt(a[1] ** 2 + a[2] ** 2 + 1))); ang_ref = min(sum(tq) / 100.0, 0.5)
        else:
            ang_ref = 0.0
        assert abs(ang - ang_ref) < 1e-9, (tick, ang, ang_ref)
        fr = rec_ref[2] + rec_ref[5] - rec_ref[8] - rec_ref[11]
        assert abs(pitch - (-ang_ref if fr > 0.05 else ang_ref)) < 1e-9


def test_swing_legs_restatement(oracle):
    """N4a oracle: stance legs track their own position (zero position error, start point refreshed), swing legs follow the quartic
    Bezier blend between start and target with 0.4 m * 6 t^2 (1-t)^2 of clearance, PD force = kp * pos error + kd * velocity error"""
    Rz = np.eye(3).reshape(9)
    foot = np.array([0.17, 0.15, -0.3, 0.17, -0.15, -0.3, -0.17, 0.15, -0.3, -0.17, -0.15, -0.3]); tgt = foot + np.tile([0.05, 0.0, 0.0], 4)
    start = np.zeros(12); rl = foot.copy(); tl = foot.copy()
    cur, kin = oracle.swing_legs(Rz, foot, [10, 10, 10, 10], tgt, start, rl, tl)
    assert (cur == foot).all() and (start == foot).all() and np.allclose(kin, 0.0, atol=1e-9)
    cur, kin = oracle.swing_legs(Rz, foot, [180, 10, 10, 180], tgt, start, rl, tl)   # legs 0, 3 at half swing: t = 0.5
    t = 0.5
    zc = 6 * t ** 2 * (1 - t) ** 2 * np.float32(0.4)
    blend = 4 * t ** 3 * (1 - t) + t ** 4 + 6 * t ** 2 * (1 - t) ** 2
    assert np.allclose(tl[0:3], [foot[0] + blend * 0.05, foot[1], foot[2] + zc], atol=1e-12)
    assert np.allclose(kin[0], 300.0 * (tl[0] - foot[0]) + 8.0 * ((tl[0] - foot[0]) / 0.0025), rtol=1e-12)


def test_leg_state_restatement(oracle):
    """N4b oracle: zero pose = the A1 geometry, Jacobian = numerical derivative of the forward kinematics, frame chain"""
    z = np.zeros(12); I = np.eye(3).reshape(9)
    o = oracle.leg_state(z, z, I, [0, 0, 0], [0, 0, 0])
    assert np.allclose(o["foot_pos_rel"].reshape(4, 3), [[0.1805, 0.047 + 0.0838, -0.42], [0.1805, -0.1308, -0.42], [-0.1805, 0.1308, -0.42], [-0.1805, -0.1308, -0.42]], atol=1e-15)
    rng = np.random.default_rng(9)
    for _ in range(50):
        q = rng.uniform(-1, 1, 12); qd = rng.normal(0, 2, 12); opt = rng.normal(0, 0.01, (4, 3))
        yaw, pit = rng.uniform(-3, 3), rng.uniform(-0.4, 0.4)
        R = (np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]) @ np.array([[np.cos(pit), 0, np.sin(pit)], [0, 1, 0], [-np.sin(pit), 0, np.cos(pit)]]))
        pos, vel = rng.normal(0, 1, 3), rng.normal(0, 1, 3)
        o = oracle.leg_state(q, qd, R.reshape(9), pos, vel, rho_opt=opt)
        J = o["Jb"].reshape(4, 3, 3).transpose(0, 2, 1)  # [leg][row][col]
        for k in range(3):
            e = np.zeros(12); h = 1e-6
            for leg in range(4):
                e[:] = 0; e[3 * leg + k] = h
                d = (oracle.leg_state(q + e, qd, R.reshape(9), pos, vel, rho_opt=opt)["foot_pos_rel"] - oracle.leg_state(q - e, qd, R.reshape(9), pos, vel, rho_opt=opt)["foot_pos_rel"]) / (2 * h)
                assert np.allclose(d[3 * leg:3 * leg + 3], J[leg][:, k], atol=1e-8)
        pr = o["foot_pos_rel"].reshape(4, 3); vr = np.einsum("lij,lj->li", J, qd.reshape(4, 3))
        assert np.allclose(o["foot_vel_rel"].reshape(4, 3), vr, atol=1e-14) and np.allclose(o["foot_pos_world"].reshape(4, 3), pr @ R.T + pos, atol=1e-14)
        assert np.allclose(o["foot_vel_world"].reshape(4, 3), vr @ R.T + vel, atol=1e-13)


def _ekf_tick_80bit(state, dt, mm, ff, R, acc, w, fk, fv, flat):
    """One tick of S/A1BasicEKF.cpp:70-147 in numpy's 80-bit long double (Gaussian elimination with partial pivoting for the two solves): the accuracy yardstick of
    the test below -- 2048x less rounding than either of the oracle's two arithmetics."""
    LD = np.longdouble
    C_ = np.zeros((28, 18), dtype=LD)
    for i in range(4):
        C_[3 * i:3 * i + 3, 0:3] = -np.eye(3); C_[3 * i:3 * i + 3, 6 + 3 * i:9 + 3 * i] = np.eye(3); C_[12 + 3 * i:15 + 3 * i, 3:6] = np.eye(3); C_[24 + i, 6 + 3 * i + 2] = 1
    x = state[:18].astype(LD); P = state[18:342].reshape(18, 18).astype(LD)
    R = R.reshape(3, 3).astype(LD); fk = fk.astype(LD); fv = fv.astype(LD); acc = acc.astype(LD); w = w.astype(LD); dt = LD(dt)
    A = np.eye(18, dtype=LD); A[0:3, 3:6] = dt * np.eye(3); B = np.zeros((18, 3), dtype=LD); B[3:6] = dt * np.eye(3)
    u = R @ acc + np.array([0, 0, LD(-9.81)], dtype=LD); e = np.ones(4, dtype=LD) if mm == 0 else np.clip(ff.astype(LD) / LD(100.0), 0, 1)
    Q = np.eye(18, dtype=LD); Q[0:3, 0:3] *= LD(0.01) * dt / 20; Q[3:6, 3:6] *= LD(0.01) * dt * LD(9.8) / 20; Rm = np.eye(28, dtype=LD)
    for i in range(4):
        k = 1 + (1 - e[i]) * LD(1e3)
        Q[6 + 3 * i:9 + 3 * i, 6 + 3 * i:9 + 3 * i] = k * dt * LD(0.01) * np.eye(3); Rm[3 * i:3 * i + 3, 3 * i:3 * i + 3] = k * LD(0.001) * np.eye(3)
        Rm[12 + 3 * i:15 + 3 * i, 12 + 3 * i:15 + 3 * i] = k * LD(0.1) * np.eye(3); Rm[24 + i, 24 + i] = k * LD(0.001) if flat else LD(1e5)
    xb = A @ x + B @ u; Pb = A @ P @ A.T + Q; y = np.zeros(28, dtype=LD)
    sk = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=LD)
    for i in range(4):
        f = fk[3 * i:3 * i + 3]; y[3 * i:3 * i + 3] = R @ f
        y[12 + 3 * i:15 + 3 * i] = (1 - e[i]) * x[3:6] + e[i] * (R @ (-fv[3 * i:3 * i + 3] - sk @ f)); y[24 + i] = (1 - e[i]) * (x[2] + f[2])
    S = C_ @ Pb @ C_.T + Rm; S = (S + S.T) / 2
    T = np.concatenate([S, (y - C_ @ xb)[:, None], C_], axis=1)          # [S | error_y | C], eliminated with partial pivoting
    for k in range(28):
        piv = k + int(np.argmax(np.abs(T[k:, k])))
        if piv != k: T[[k, piv]] = T[[piv, k]]
        for i in range(k + 1, 28): T[i, k:] -= (T[i, k] / T[k, k]) * T[k, k:]
    X = np.zeros((28, 19), dtype=LD)
    for k in range(27, -1, -1): X[k] = (T[k, 28:] - T[k, k + 1:28] @ X[k + 1:]) / T[k, k]
    xn = xb + Pb @ C_.T @ X[:, 0]; Pn = Pb - Pb @ C_.T @ X[:, 1:] @ Pb; Pn = (Pn + Pn.T) / 2
    if Pn[0, 0] * Pn[1, 1] - Pn[0, 1] * Pn[1, 0] > 1e-6:
        Pn[0:2, 2:] = 0; Pn[2:, 0:2] = 0; Pn[0:2, 0:2] /= 10
    return xn, Pn


def test_ekf_device_variant_vs_the_pinned_restatement_and_an_80_bit_evaluation(oracle, scen):
    """orc_ekf_step is the pinned restatement (multiply + add, the two solves as products with an explicit S^-1; held to the reference's compiled source by
    tests/test_ref_pin.py); orc_ekf_step_device is the device kernel's arithmetic (round 6: L D L' of S with [C Pbar | error_y] riding along, no S^-1) and is no ground
    truth of its own.  Two bounds:
      (i)  over 20 robots x 200 ticks the two stay within 1e-10 m / m/s of each other (measured 2.2e-11);
      (ii) tick by tick from the same state, against an 80-bit evaluation of the reference's formulas, the device arithmetic is the CLOSER of the two: state within
           1e-13 (measured 9.8e-15; the pinned variant 7.3e-14), covariance within 1e-13 (measured 1.0e-15; pinned 1.0e-12) -- what separates the two variants in
           (i) is the explicit inverse's rounding, not the elimination's."""
    rng = np.random.default_rng(51)
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    worst = 0.0
    ex = [0.0, 0.0]; eP = [0.0, 0.0]
    for rob in range(20):
        s0 = oracle.ekf_state(); s1 = oracle.ekf_state()
        for t in range(200):
            mm = 1 if (t > 3 and rng.random() < 0.8) else 0
            e = rng.normal(0, 0.05, 2); R = scen.rot_zyx(e[0], e[1], rng.uniform(-3, 3)).reshape(9)
            fk = base + rng.normal(0, 0.01, 12); fv = rng.normal(0, 0.3, 12); acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.3, 3); w = rng.normal(0, 0.3, 3); ff = rng.uniform(0, 160, 4)
            yard = rob < 4 and 0 < t <= 40
            if yard:   # both arithmetics from ONE state (the pinned sequence's), against the 80-bit tick
                xt, Pt = _ekf_tick_80bit(s0, 0.0025, mm, ff, R, acc, w, fk, fv, rob % 2)
                sd = s0.copy()
                oracle.ekf_step(sd, 0.0025, mm, ff, R, acc, w, fk, fv, assume_flat_ground=rob % 2, device=True)
            p0, v0, e0 = oracle.ekf_step(s0, 0.0025, mm, ff, R, acc, w, fk, fv, assume_flat_ground=rob % 2)
            p1, v1, e1 = oracle.ekf_step(s1, 0.0025, mm, ff, R, acc, w, fk, fv, assume_flat_ground=rob % 2, device=True)
            if yard:
                for i, s_ in enumerate((s0, sd)):
                    ex[i] = max(ex[i], float(np.abs(s_[:18] - xt).max())); eP[i] = max(eP[i], float(np.abs(s_[18:342].reshape(18, 18) - Pt).max()))
            worst = max(worst, np.abs(p0 - p1).max(), np.abs(v0 - v1).max())
            assert (e0 == e1).all()
    assert 0.0 < worst <= 1e-10, worst   # (> 0: the two variants really are different arithmetics)
    assert ex[1] <= 1e-13 and eP[1] <= 1e-13, (ex, eP)
    assert ex[1] <= ex[0] and eP[1] <= eP[0], (ex, eP)


def test_ekf_restatement(oracle):
    """N4c oracle vs an independent numpy Kalman update (numpy.linalg.solve for the two S^-1 products) over a 120-tick sequence"""
    rng = np.random.default_rng(13)
    C_ = np.zeros((28, 18))
    for i in range(4):
        C_[3 * i:3 * i + 3, 0:3] = -np.eye(3); C_[3 * i:3 * i + 3, 6 + 3 * i:9 + 3 * i] = np.eye(3); C_[12 + 3 * i:15 + 3 * i, 3:6] = np.eye(3); C_[24 + i, 6 + 3 * i + 2] = 1
    st = oracle.ekf_state(); x = None; P = None
    for t in range(120):
        dt = 0.0025; mm = 1 if t > 5 else 0
        yaw = 0.3 * np.sin(t / 30); R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
        fk = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3]) + rng.normal(0, 0.01, 12)
        fv = rng.normal(0, 0.2, 12); acc = np.array([0.1, -0.2, 9.81]) + rng.normal(0, 0.05, 3); w = rng.normal(0, 0.2, 3); ff = rng.uniform(0, 150, 4)
        pos, vel, ec = oracle.ekf_step(st, dt, mm, ff, R.reshape(9), acc, w, fk, fv)
        if x is None:
            x = np.zeros(18); x[2] = 0.09; P = 3 * np.eye(18)
            for i in range(4): x[6 + 3 * i:9 + 3 * i] = R @ fk[3 * i:3 * i + 3] + x[0:3]
        else:
            A = np.eye(18); A[0:3, 3:6] = dt * np.eye(3); B = np.zeros((18, 3)); B[3:6] = dt * np.eye(3)
            u = R @ acc + [0, 0, -9.81]; e = np.ones(4) if mm == 0 else np.clip(ff / 100.0, 0, 1)
            Q = np.eye(18); Q[0:3, 0:3] *= 0.01 * dt / 20; Q[3:6, 3:6] *= 0.01 * dt * 9.8 / 20; Rm = np.eye(28)
            for i in range(4):
                k = 1 + (1 - e[i]) * 1e3
                Q[6 + 3 * i:9 + 3 * i, 6 + 3 * i:9 + 3 * i] = k * dt * 0.01 * np.eye(3); Rm[3 * i:3 * i + 3, 3 * i:3 * i + 3] = k * 0.001 * np.eye(3)
                Rm[12 + 3 * i:15 + 3 * i, 12 + 3 * i:15 + 3 * i] = k * 0.1 * np.eye(3); Rm[24 + i, 24 + i] = k * 0.001
            xb = A @ x + B @ u; Pb = A @ P @ A.T + Q; y = np.zeros(28)
            sk = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            for i in range(4):
                f = fk[3 * i:3 * i + 3]; y[3 * i:3 * i + 3] = R @ f
                y[12 + 3 * i:15 + 3 * i] = (1 - e[i]) * x[3:6] + e[i] * (R @ (-fv[3 * i:3 * i + 3] - sk @ f)); y[24 + i] = (1 - e[i]) * (x[2] + f[2])
            S = C_ @ Pb @ C_.T + Rm; S = 0.5 * (S + S.T)
            x = xb + Pb @ C_.T @ np.linalg.solve(S, y - C_ @ xb); P = Pb - Pb @ C_.T @ np.linalg.solve(S, C_) @ Pb; P = 0.5 * (P + P.T)
            if np.linalg.det(P[0:2, 0:2]) > 1e-6:
                P[0:2, 2:] = 0; P[2:, 0:2] = 0; P[0:2, 0:2] /= 10
            assert (ec == (e >= 0.5)).all()
        assert np.allclose(pos, x[0:3], atol=1e-10) and np.allclose(vel, x[3:6], atol=1e-10), (t, pos, x[0:3])
        assert np.allclose(st[18:18 + 324].reshape(18, 18), P, rtol=1e-8, atol=1e-12)


def test_two_linear_system_back_ends_agree(oracle, scen):
    """settings.linsys: reduced dense Cholesky vs LDL' of the full quasi-definite KKT matrix (what OSQP's QDLDL factors): same iteration
    count, status and factorisation count on every QP, forces within 1e-5 N (observed ~1e-6: the resolution of any "same as OSQP" claim).
    The 1e5-QP soak is tests/tools/linsys_soak.py -> profiles/r02_linsys_soak.json."""
    from helpers import oracle_batch
    for gen, kw in (("config3_random_flat", dict(nb=48)), ("config4_random_h16", dict(nb=8)), ("config5_divergent", dict(nb=8))):
        sc = getattr(scen, gen)(**kw)
        a = oracle_batch(oracle, sc); b = oracle_batch(oracle, sc, settings=oracle.default_settings(linsys=1))
        assert (a["iters"] == b["iters"]).all() and (a["status"] == b["status"]).all() and (a["nfact"] == b["nfact"]).all()
        assert np.abs(a["u"] - b["u"]).max() <= 1e-5


def test_update_path_restatement(oracle, scen):
    """The reference's tick >= 2 path (updateHessianMatrix / updateGradient / update*Bound on the persistent OsqpEigen workspace,
    S/A1RobotControl.cpp:533-538) restated from OSQP 0.6's update functions (osqp_solve_impl with a carry): mechanics only --
    the first tick is the cold solve bit for bit; a tick that repeats the previous tick's QP re-derives the same scaling, starts at the previous
    solution and stops at the first check with (nearly) the same forces; a contact switch changes constraint types and costs a
    factorisation; a failed tick leaves a usable workspace behind; over a slowly moving trot the path stays within OSQP's own slack of the
    "fresh set-up + osqp_warm_start" restatement the engine implements (tests/tools/update_path_probe.py has the 4000-tick figures)."""
    seq = scen.config2_trot_sequence(130)
    p = seq["params"]
    pr = oracle.mpc_params(10, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    args = lambda k: (seq["x0"][k], seq["xref"][k], seq["R"][k], seq["foot"][k], seq["contact"][k])
    carry = oracle.update_carry(10)
    a = oracle.mpc_solve_update(pr, st, *args(0), carry)
    cold = oracle.mpc_solve(pr, oracle.default_settings(warm_start=0), *args(0))
    assert np.array_equal(a["grf"], cold["grf"]) and a["info"].iters == cold["info"].iters and carry[0] == 1.0
    b = oracle.mpc_solve_update(pr, st, *args(0), carry)          # the same QP again: same scaling, start = the previous solution
    opt = oracle.mpc_solve(pr, oracle.exact_settings(), *args(0))["grf"]   # (a default-tolerance answer is N's away from the optimum: 25 more iterations move it closer)
    assert b["info"].iters == 25 and b["info"].nfact == 1 and np.abs(b["grf"] - opt).max() < np.abs(a["grf"] - opt).max()
    # contact switch at tick 60 (1001 -> 0110): eight rows change type, OSQP refactors once more inside updateUpperBound
    for k in range(1, 60):
        r = oracle.mpc_solve_update(pr, st, *args(k), carry)
    assert r["info"].status in (1, 2)
    sw = oracle.mpc_solve_update(pr, st, *args(60), carry)
    assert sw["info"].nfact >= 2 and sw["info"].status in (1, 2) and np.abs(sw["grf"].reshape(4, 3)[[0, 3]]).max() < 0.5   # legs 0 and 3 now swing (to the solver's tolerance)
    # a non-finite tick: zeros out, cold iterates left in the workspace, the next tick is solved
    bad = seq["x0"][61].copy(); bad[3] = np.nan
    nf = oracle.mpc_solve_update(pr, st, bad, *args(61)[1:], carry)
    assert nf["info"].status == -7 and not nf["grf"].any() and not carry[2:2 + 120].any()
    ok = oracle.mpc_solve_update(pr, st, *args(62), carry)
    assert ok["info"].status in (1, 2) and np.isfinite(ok["grf"]).all()
    # distance to the engine's semantics over a slowly moving state (same QP data tick by tick, two warm starts of it)
    rng = np.random.default_rng(3)
    x0 = seq["x0"][0].copy(); wx = np.zeros(120); wy = np.zeros(200); rho = 0.0; carry = oracle.update_carry(10); d = []; same = 0
    for k in range(120):
        x0[:12] += rng.normal(0, 5e-4, 12)
        fresh = oracle.mpc_solve(pr, st, x0, seq["xref"][0], seq["R"][0], seq["foot"][0], seq["contact"][0], warm_x=wx, warm_y=wy, warm_rho=rho)
        wx, wy, rho = fresh["warm_x"], fresh["warm_y"], fresh["rho"]
        upd = oracle.mpc_solve_update(pr, st, x0, seq["xref"][0], seq["R"][0], seq["foot"][0], seq["contact"][0], carry)
        d.append(np.abs(fresh["grf"] - upd["grf"]).max()); same += fresh["info"].iters == upd["info"].iters
    assert d[0] == 0.0 and np.median(d) < 1.0 and max(d) < 20.0 and same >= 100, (np.median(d), max(d), same)


def test_check_termination_reproduces_osqp_residuals(oracle, scen):
    """orc_check_termination (round 4: OSQP's termination test on a GIVEN unscaled iterate, used by the GPU suite on the ticks where engine and oracle part) evaluated on
    the oracle's OWN solution (x, y from the solve, z from orc_last_z) reproduces the residuals the solve itself reported and passes; a perturbed point fails."""
    sc = scen.config3_random_flat(nb=6)
    p = sc["params"]; pr = oracle.mpc_params(10, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    for i in range(6):
        P, g, _, _, _, csr = oracle.mpc_form(pr, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i])
        r = oracle.mpc_solve(pr, st, sc["x0"][i], sc["xref"][i], sc["R"][i], sc["foot"][i], sc["contact"][i], warm_x=np.zeros(120), warm_y=np.zeros(200))
        z = oracle.last_z(200)
        c = oracle.check_termination(P, g, csr, r["warm_x"], z, r["warm_y"])
        assert r["info"].status == 1 and c["ok"]
        assert abs(c["pri_res"] - r["info"].pri_res) <= 1e-12 * max(1.0, r["info"].pri_res) and abs(c["dua_res"] - r["info"].dua_res) <= 1e-12
        bad = oracle.check_termination(P, g, csr, r["warm_x"] + 5.0, z, r["warm_y"])
        assert not bad["ok"]
