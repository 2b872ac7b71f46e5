"""GPU, through the C ABI: the caller-side rows of SURVEY 8(f) -- gait plan, contacts / terrain, swing legs, leg kinematics, EKF, joint torques, the control tick in one call -- bit-exact vs the oracle (which oracle/_ref pins to the reference's sources)."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import *  # noqa: F401,F403  (_engine, _strided_inputs, tick_inputs, TICK_STATE, _oracle_update_ticks, SETTINGS_CASES)
from gpu_common import _engine, _oracle_update_ticks, _strided_inputs  # noqa: F401
from helpers import TOL_FORCE_BALANCE_N, TOL_FORCE_N, compare, exact_resolver, noise_band, oracle_batch, oracle_params, take  # noqa: F401

pytestmark = pytest.mark.gpu


def test_update_plan_N2a_bit_exact(pkg, oracle, scen):
    """gait counters, planned contacts, Raibert foothold (S/A1RobotControl.cpp:148-202) -- element-wise arithmetic,
    so the bar is BIT-exact against the oracle's restatement (both built without FMA contraction)."""
    rng = np.random.default_rng(7)
    n = 5000
    yaw = rng.uniform(-np.pi, np.pi, n); roll = rng.uniform(-0.2, 0.2, n); pit = rng.uniform(-0.2, 0.2, n)
    R = scen.rot_zyx(roll, pit, yaw).reshape(n, 9); Rz = scen.rot_zyx(0 * yaw, 0 * yaw, yaw).reshape(n, 9)
    mm = (rng.random(n) < 0.8).astype(np.uint8)
    gc = rng.uniform(0, 240, (n, 4)); gc[::7] = [0, 120, 120, 0]; gc[::11, 0] = 239.0  # wrap-around through fmod
    spd = rng.choice([1.0, 1.5, 2.0, 3.0], size=(n, 4))
    v = rng.normal(0, 0.6, (n, 3)); vd = rng.normal(0, 0.6, (n, 3)); vd[::5] *= 10  # saturates FOOT_DELTA_*_LIMIT
    pos = rng.normal(0, 2.0, (n, 3))
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.update_plan(mm, gc, spd, v, Rz, R, pos, vd)
    gp = oracle.gait_params([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35])
    for b in range(0, n, 3):
        g2, pc, rel, ab, wo = oracle.update_plan(gp, mm[b], gc[b], spd[b], v[b], Rz[b], R[b], pos[b], vd[b])
        assert (out["gait_counter"][b] == g2).all() and (out["plan_contacts"][b] == pc).all()
        assert (out["foot_pos_target_rel"][b] == rel).all() and (out["foot_pos_target_abs"][b] == ab).all()
        assert (out["foot_pos_target_world"][b] == wo).all()
    assert (out["plan_contacts"][mm == 0] == 1).all() and (np.abs(out["foot_pos_target_rel"].reshape(n, 4, 3)[:, :, 0] - [0.17, 0.17,
            -0.17, -0.17]) <= 0.1 + 1e-15).all()


def test_joint_torques_N3_bit_exact(pkg, oracle, scen):
    """tau = J'(-f) on stance legs, J^-1 (km .* f_kin) by partial-pivot LU on swing legs, gravity term, NaN guard
    (S/A1RobotControl.cpp:289-319): bit-exact against the oracle's restatement."""
    rng = np.random.default_rng(11)
    n = 4000
    Jb = rng.normal(0, 0.2, (n, 4, 9)); Jb[:, :, [0, 4, 8]] += rng.choice([-0.3, 0.3], size=(n, 4, 3))  # every pivot pattern occurs
    Jb[5, 1] = 0.0  # singular block -> NaN -> previous torque kept
    c = (rng.random((n, 4)) < 0.5).astype(np.uint8); act = (rng.random(n) < 0.9).astype(np.uint8)
    grf = rng.normal(0, 40, (n, 12)); fk = rng.normal(0, 20, (n, 12)); tg = rng.normal(0, 1, (n, 12)); prev = rng.normal(0, 5, (n, 12))
    km = np.array([0.1, 0.1, 0.04])
    c[5, 1] = 0; act[5] = 1
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, n, 0) as eng:
        tau = eng.joint_torques(act, c, Jb.reshape(n, 36), grf, fk, km, tg, prev)
    for b in range(0, n, 2):
        ref = oracle.joint_torques(act[b], c[b], Jb[b].reshape(36), grf[b], fk[b], km, tg[b], prev[b])
        assert (tau[b] == ref).all(), b
    ref5 = oracle.joint_torques(1, c[5], Jb[5].reshape(36), grf[5], fk[5], km, tg[5], prev[5])
    assert (tau[act == 0] == 0).all() and np.array_equal(tau[5], ref5)
    # 0*inf = NaN is guarded (:314-317), the infinity is not -- like the reference
    assert (tau[5, 3:5] == prev[5, 3:5]).all() and np.isinf(tau[5, 5])


def test_contact_terrain_N2b_sequence(pkg, oracle, scen):
    """150 ticks of contact logic + moving-window filters + plane fit + terrain pitch for 300 robots, device-resident
    filter state vs the oracle's per-robot state (S/A1RobotControl.cpp:256-282, 566-582, 335-376).  Contacts and the filtered contact
    positions are bit-exact (same arithmetic, no contraction); the terrain angle goes through acos (device math library vs glibc)."""
    rng = np.random.default_rng(21)
    n, ticks = 300, 150
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    states = [oracle.contact_state() for _ in range(n)]
    pitch_g = np.zeros(n); pitch_o = np.zeros(n)
    base = np.outer([0.2, 0.2, -0.2, -0.2], [1.0, 0.0, 0.3]).reshape(12) + np.outer([1, -1, 1, -1], [0.0, 0.13, 0.0]).reshape(12)
    gcs = rng.uniform(0, 240, (n, 4))
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            gcs = np.fmod(gcs + 2.0, 240.0)
            plan = (gcs <= 120).astype(np.uint8)
            ff = rng.uniform(0, 80, (n, 4))
            foot = base + rng.normal(0, 0.03, (n, 12)) + np.tile([0.0, 0.0, -0.3], 4)
            z = np.where(rng.random(n) < 0.9, 0.3, 0.05)
            out = eng.contact_terrain(gcs, plan, ff, foot, z, pitch_g)
            pitch_g = out["root_euler_d_pitch"]
            for b in range(0, n, 7):
                ct, rec, ang, pitch_o[b] = oracle.contact_terrain_step(states[b], gcs[b], plan[b], ff[b], foot[b], z[b], pitch_o[b])
                assert (out["contacts"][b] == ct).all() and (out["foot_pos_recent_contact"][b] == rec).all(), (t, b)
                assert abs(out["terrain_angle"][b] - ang) <= 1e-13 and abs(pitch_g[b] - pitch_o[b]) <= 1e-13, (t, b)
        eng.reset_contact_state()
        out = eng.contact_terrain(gcs, plan, ff, foot, z, np.zeros(n))
        fresh = oracle.contact_state()
        ct, rec, ang, _ = oracle.contact_terrain_step(fresh, gcs[0], plan[0], ff[0], foot[0], z[0], 0.0)
        assert (out["foot_pos_recent_contact"][0] == rec).all() and abs(out["terrain_angle"][0] - ang) <= 1e-13


def test_contact_terrain_N2b_partial_wavefront_and_spare_capacity(pkg, oracle, scen):
    """N2b with n = 70 robots on a handle created for 200: the second wavefront of the launch stages 6 records (the record copy is cut at
    n, the ring regions start
    behind max_batch records) -- every robot against the oracle over a leg-window wrap."""
    rng = np.random.default_rng(77)
    n, cap, ticks = 70, 200, 90
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    states = [oracle.contact_state() for _ in range(n)]
    pitch_g = np.zeros(n); pitch_o = np.zeros(n)
    gcs = rng.uniform(0, 240, (n, 4))
    with pkg.Engine(cfg, cap, 0) as eng:
        for t in range(ticks):
            gcs = np.fmod(gcs + 2.0, 240.0)
            plan = (gcs <= 150).astype(np.uint8); ff = rng.uniform(0, 80, (n, 4)); foot = rng.normal(0, 0.2, (n, 12)); z = np.full(n, 0.3)
            out = eng.contact_terrain(gcs, plan, ff, foot, z, pitch_g); pitch_g = out["root_euler_d_pitch"]
            for b in range(n):
                ct, rec, ang, pitch_o[b] = oracle.contact_terrain_step(states[b], gcs[b], plan[b], ff[b], foot[b], z[b], pitch_o[b])
                assert (out["contacts"][b] == ct).all() and (out["foot_pos_recent_contact"][b] == rec).all(), (t, b)
                assert abs(out["terrain_angle"][b] - ang) <= 1e-13 and abs(pitch_g[b] - pitch_o[b]) <= 1e-13, (t, b)


def test_swing_legs_N4a_sequence(pkg, oracle, scen):
    """swing-leg Bezier targets + foot PD force over 60 ticks for 500 robots (S/A1RobotControl.cpp:204-254).  The carried
    state and foot_pos_cur are bit-exact; the curve uses products for the integer powers (std::pow in the reference): a few ulp."""
    rng = np.random.default_rng(31)
    n, ticks = 500, 60
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    st_g = [np.zeros((n, 12)) for _ in range(3)]; st_o = [np.zeros((n, 12)) for _ in range(3)]
    gcs = rng.uniform(0, 240, (n, 4))
    base = np.array([0.17, 0.15, -0.3, 0.17, -0.15, -0.3, -0.17, 0.15, -0.3, -0.17, -0.15, -0.3])
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            gcs = np.fmod(gcs + 2.0, 240.0)
            yaw = rng.uniform(-3, 3, n); Rz = scen.rot_zyx(0 * yaw, 0 * yaw, yaw).reshape(n, 9)
            foot = base + rng.normal(0, 0.03, (n, 12)); tgt = base + rng.normal(0, 0.05, (n, 12))
            cur, kin = eng.swing_legs(Rz, foot, gcs, tgt, *st_g)
            for b in range(0, n, 11):
                c_o, k_o = oracle.swing_legs(Rz[b], foot[b], gcs[b], tgt[b], st_o[0][b], st_o[1][b], st_o[2][b])
                assert (cur[b] == c_o).all() and (st_g[0][b] == st_o[0][b]).all() and (st_g[1][b] == st_o[1][b]).all(), (t, b)
                assert np.abs(st_g[2][b] - st_o[2][b]).max() <= 1e-15 and np.abs(kin[b] - k_o).max() <= 1e-9, (t, b,
                        np.abs(kin[b] - k_o).max())
                st_o[2][b] = st_g[2][b]  # keep the two state copies from drifting apart by the curve's ulp differences


def test_control_tick_chain(pkg, oracle, scen):
    """The caller-side rows composed the way the reference's 400 Hz loop composes them (S/A1RobotControl.cpp: update_plan ->
    generate_swing_legs_ctrl -> compute_grf [terrain pitch -> MPC] -> compute_joint_torques), 48 robots x 16 ticks with synthetic
    sensor inputs, device entry points vs the oracle functions chained the same way: the joint torques at the end of every tick."""
    rng = np.random.default_rng(77)
    n, ticks, h = 48, 16, 10
    P = scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS
    cfg = pkg.make_config(P, h, warm_start=0)
    pr = oracle.mpc_params(h, P["dt"], P["mu"], P["fz_min"], P["fz_max"], P["q"], P["r"], P["mass"],
            P["inertia"]); st = oracle.default_settings()
    dfp = np.array([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35]); gp = oracle.gait_params(dfp)
    km = np.array([0.1, 0.1, 0.04])
    G = dict(gc=np.tile([0.0, 120.0, 120.0, 0.0], (n, 1)), start=np.zeros((n, 12)), rl=np.tile(dfp, (n, 1)), tl=np.tile(dfp, (n, 1)),
            pitch=np.zeros(n), tau=np.zeros((n, 12)))
    O = dict(gc=G["gc"].copy(), start=np.zeros((n, 12)), rl=G["rl"].copy(), tl=G["tl"].copy(), pitch=np.zeros(n), tau=np.zeros((n, 12)),
             ct=[oracle.contact_state() for _ in range(n)])
    worst = 0.0
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            # synthetic sensors of this tick
            eul = rng.normal(0, 0.05, (n, 3)); eul[:, 2] = rng.uniform(-1, 1, n); pos = np.c_[rng.normal(0, 1, (n, 2)),
                    0.3 + rng.normal(0, 0.01, n)]
            w = rng.normal(0, 0.3, (n, 3)); v = rng.normal(0, 0.3, (n, 3)); vd = np.c_[rng.uniform(-0.5, 0.5, (n, 2)),
                    np.zeros(n)]; wd = np.c_[np.zeros((n, 2)), rng.uniform(-0.5, 0.5, n)]
            R = scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9); Rz = scen.rot_zyx(0 * eul[:, 0], 0 * eul[:, 0],
                    eul[:, 2]).reshape(n, 9)
            foot_rel = dfp + rng.normal(0, 0.02, (n, 12))
            foot_abs = np.einsum("nij,nlj->nli", R.reshape(n, 3, 3), foot_rel.reshape(n, 4, 3)).reshape(n, 12)
            ff = rng.uniform(0, 80, (n, 4)); Jb = rng.normal(0, 0.2, (n, 36)); Jb[:, [0, 4, 8, 9, 13, 17, 18, 22, 26, 27, 31, 35]] += 0.3
            tg = rng.normal(0, 0.5, (n, 12)); mm = np.ones(n, np.uint8); spd = np.full((n, 4), 2.0)
            # ---- device chain
            up = eng.update_plan(mm, G["gc"], spd, v, Rz, R, pos, vd); G["gc"] = up["gait_counter"]
            cur, kin = eng.swing_legs(Rz, foot_abs, G["gc"], up["foot_pos_target_rel"], G["start"], G["rl"], G["tl"])
            ctr = eng.contact_terrain(G["gc"], up["plan_contacts"], ff, foot_abs, pos[:, 2],
                    G["pitch"]); G["pitch"] = ctr["root_euler_d_pitch"]
            eul_d = np.c_[np.zeros(n), G["pitch"], eul[:, 2]]
            tick = scen.pack_tick(eul, pos, w, v, eul_d, vd, wd, np.full(n, 0.3))
            sol = eng.solve_ticks(tick, R, foot_abs, ctr["contacts"])
            G["tau"] = eng.joint_torques(np.ones(n, np.uint8), ctr["contacts"], Jb, sol["grf"], kin, km, tg, G["tau"])
            # ---- oracle chain
            for b in range(n):
                gc2, pc, rel, ab, wo = oracle.update_plan(gp, 1, O["gc"][b], spd[b], v[b], Rz[b], R[b], pos[b], vd[b]); O["gc"][b] = gc2
                c_o, k_o = oracle.swing_legs(Rz[b], foot_abs[b], gc2, rel, O["start"][b], O["rl"][b], O["tl"][b])
                ct, rec, ang, O["pitch"][b] = oracle.contact_terrain_step(O["ct"][b], gc2, pc, ff[b], foot_abs[b], pos[b, 2], O["pitch"][b])
                ed = np.array([0.0, O["pitch"][b], eul[b, 2]])
                xref = oracle.mpc_reference(h, P["dt"], eul[b], pos[b], R[b], ed, vd[b], wd[b], 0.3)
                x0 = scen.pack_x0(eul[b:b + 1], pos[b:b + 1], w[b:b + 1], v[b:b + 1])[0]
                grf = oracle.mpc_solve(pr, st, x0, xref, R[b], foot_abs[b], ct)["grf"]
                O["tau"][b] = oracle.joint_torques(1, ct, Jb[b], grf, k_o, km, tg[b], O["tau"][b])
                assert (ctr["contacts"][b] == ct).all() and (G["gc"][b] == gc2).all()
            worst = max(worst, np.abs(G["tau"] - O["tau"]).max())
            O["tl"][:] = G["tl"]  # the curve's ulp differences must not accumulate into the comparison (see test_swing_legs_N4a_sequence)
    assert worst < 1e-5, worst


def test_leg_state_N4b(pkg, oracle, scen):
    """leg forward kinematics, Jacobians and the frame chain of the joint-state callback (S/GazeboA1ROS.cpp:264-279) for
    3000 robots vs the oracle; sin / cos come from the device math library, so the bar is a few ulp of the 0.4 m leg (1e-14)."""
    rng = np.random.default_rng(41)
    n = 3000
    q = rng.uniform(-1.2, 1.2, (n, 12)); qd = rng.normal(0, 3, (n, 12)); opt = rng.normal(0, 0.01, (4, 3))
    eul = rng.uniform(-0.5, 0.5, (n, 3)); eul[:, 2] = rng.uniform(-3, 3, n); R = scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9)
    pos = rng.normal(0, 2, (n, 3)); vel = rng.normal(0, 1, (n, 3))
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    with pkg.Engine(cfg, n, 0) as eng:
        out = eng.leg_state(q, qd, R, pos, vel, rho_opt=opt)
    for b in range(0, n, 5):
        ref = oracle.leg_state(q[b], qd[b], R[b], pos[b], vel[b], rho_opt=opt)
        for k, tol in (("foot_pos_rel", 1e-14), ("Jb", 1e-14), ("foot_vel_rel", 1e-13), ("foot_pos_abs", 1e-14), ("foot_vel_abs", 1e-13),
                       ("foot_pos_world", 1e-14), ("foot_vel_world", 1e-13)):
            assert np.abs(out[k][b] - ref[k]).max() <= tol, (b, k, np.abs(out[k][b] - ref[k]).max())


def test_ekf_N4c_sequence(pkg, oracle, scen):
    """A1BasicEKF for 200 robots over 80 ticks, device-resident filter state vs the oracle's dense restatement
    (S/A1BasicEKF.cpp:54-163).  Two checks (the oracle must not move with the kernel): (i) against the PINNED restatement --
    multiply + add, the two solves as products with an explicit S^-1, what tests/test_ref_pin.py holds to the reference's compiled source --
    within 1e-10: the kernel (round 6) eliminates [S | C Pbar | error_y] to L D L' form and never forms S^-1; the CPU suite measures 2.2e-11
    between the two arithmetics over 200 ticks and shows, against an 80-bit evaluation, that it is the explicit inverse's rounding
    (tests/test_oracle.py); (ii) bit for bit against the oracle's device variant (same operations in the same order: what pins the kernel's
    lane map and elimination)."""
    rng = np.random.default_rng(51)
    n, ticks = 200, 80
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    states = [oracle.ekf_state() for _ in range(n)]; pinned = [oracle.ekf_state() for _ in range(n)]
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            mm = np.where(rng.random(n) < 0.8, 1, 0).astype(np.uint8) if t > 3 else np.zeros(n, np.uint8)
            yaw = rng.uniform(-3, 3, n); eul = rng.normal(0, 0.05, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9)
            fk = base + rng.normal(0, 0.01, (n, 12)); fv = rng.normal(0, 0.3, (n, 12)); acc = np.array([0.0, 0.0, 9.81]) + rng.normal(0,
                    0.3, (n, 3))
            w = rng.normal(0, 0.3, (n, 3)); ff = rng.uniform(0, 160, (n, 4))
            pos, vel, ec = eng.ekf_update(0.0025, mm, ff, R, acc, w, fk, fv)
            for b in range(0, n, 3):
                p_o, v_o, e_o = oracle.ekf_step(states[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], device=True)
                assert np.array_equal(pos[b], p_o) and np.array_equal(vel[b], v_o) and (ec[b] == e_o).all(), (t, b, pos[b] - p_o,
                        vel[b] - v_o)
                p_p, v_p, e_p = oracle.ekf_step(pinned[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b])
                assert max(np.abs(pos[b] - p_p).max(), np.abs(vel[b] - v_p).max()) <= 1e-10 and (ec[b] == e_p).all(), (t, b, pos[b] - p_p,
                        vel[b] - v_p)


def test_ekf_large_batch_residency(pkg, oracle, scen):
    """A batch of several rounds of the chip (16 385 robots: the last workgroup half empty), five ticks from the first-call
    initialisation on: a spread of robots incl. the first and the last against the oracle's device variant bit for bit, and the robots of a
    200-robot engine against the same rows of the large batch bit for bit (a robot's result does not depend on the batch it rides in)."""
    rng = np.random.default_rng(52)
    n, ticks, small = 16385, 5, 200
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    sample = sorted({0, 1, 2, small - 1, n // 2, n - 2, n - 1} | set(rng.integers(0, n, 40).tolist()))
    states = {b: oracle.ekf_state() for b in sample}
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    with pkg.Engine(cfg, n, 0) as eng, pkg.Engine(cfg, small, 0) as eng_s:
        for t in range(ticks):
            mm = np.where(rng.random(n) < 0.8, 1, 0).astype(np.uint8) if t > 1 else np.zeros(n, np.uint8)
            yaw = rng.uniform(-3, 3, n); eul = rng.normal(0, 0.05, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9)
            fk = base + rng.normal(0, 0.01, (n, 12)); fv = rng.normal(0, 0.3, (n, 12))
            acc = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.3, (n, 3))
            w = rng.normal(0, 0.3, (n, 3)); ff = rng.uniform(0, 160, (n, 4))
            pos, vel, ec = eng.ekf_update(0.0025, mm, ff, R, acc, w, fk, fv)
            s = slice(0, small)
            pos_s, vel_s, ec_s = eng_s.ekf_update(0.0025, mm[s], ff[s], R[s], acc[s], w[s], fk[s], fv[s])
            assert np.array_equal(pos[s], pos_s) and np.array_equal(vel[s], vel_s) and np.array_equal(ec[s], ec_s), t
            for b in sample:
                p_o, v_o, e_o = oracle.ekf_step(states[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], device=True)
                assert np.array_equal(pos[b], p_o) and np.array_equal(vel[b], v_o) and (ec[b] == e_o).all(), (t, b, pos[b] - p_o, vel[b] - v_o)


def test_device_pointer_tick_matches_host_pointer_tick(pkg, scen):
    """The *_device variants of the caller-side entry points chained on the GPU (torch tensors, one stream, no host copies between the
    stages) give bit for bit what the host-pointer entries give: leg state -> EKF -> plan -> swing legs -> contacts / terrain -> MPC
    (tick records) -> joint torques, three ticks, 256 robots."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(99)
    n, h = 256, 10
    P = scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS
    cfg = pkg.make_config(P, h, warm_start=0)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    km = np.array([0.1, 0.1, 0.04]); kp = np.array([300.0, 400.0, 400.0]); kd = np.array([8.0, 8.0,
            8.0]); dp_ = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    with pkg.Engine(cfg, n, 0) as eh, pkg.Engine(cfg, n, 0) as ed:
        gait = pkg.engine.GaitConfig(); ed.lib.a1mpc_default_gait_config(C.byref(gait)); ccfg = pkg.engine.ContactConfig(
                ); ed.lib.a1mpc_default_contact_config(C.byref(ccfg))
        fix = np.ascontiguousarray(eh.A1_RHO_FIX); opt = np.zeros((4, 3))
        st_h = dict(gc=np.tile([0.0, 120.0, 120.0, 0.0], (n, 1)), start=np.zeros((n, 12)), rl=np.zeros((n, 12)), tl=np.zeros((n, 12)),
                pitch=np.zeros(n), tau=np.zeros((n, 12)))
        st_d = {k: T(v) for k, v in st_h.items()}
        st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
        for t in range(3):
            q = rng.uniform(-0.8, 0.8, (n, 12)); qd = rng.normal(0, 1, (n, 12)); eul = rng.normal(0, 0.05, (n, 3)); eul[:,
                    2] = rng.uniform(-1, 1, n)
            R = scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9); Rz = scen.rot_zyx(0 * eul[:, 0], 0 * eul[:, 0],
                    eul[:, 2]).reshape(n, 9)
            acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.2, (n, 3)); w = rng.normal(0, 0.2, (n, 3)); ff = rng.uniform(0, 120,
                    (n, 4)); mm = np.ones(n, np.uint8)
            vd = np.c_[rng.uniform(-0.4, 0.4, (n, 2)), np.zeros(n)]; wd = np.c_[np.zeros((n, 2)),
                    rng.uniform(-0.4, 0.4, n)]; spd = np.full((n, 4), 2.0); tg = rng.normal(0, 0.5, (n, 12))
            act = np.ones(n, np.uint8)
            # ---- host-pointer chain
            leg = eh.leg_state(q, qd, R, np.zeros((n, 3)), np.zeros((n, 3)))
            pos, vel, ec = eh.ekf_update(0.0025, mm, ff, R, acc, w, leg["foot_pos_rel"], leg["foot_vel_rel"])
            up = eh.update_plan(mm, st_h["gc"], spd, vel, Rz, R, pos, vd); st_h["gc"] = up["gait_counter"]
            cur, kin = eh.swing_legs(Rz, leg["foot_pos_abs"], st_h["gc"], up["foot_pos_target_rel"], st_h["start"], st_h["rl"], st_h["tl"])
            ctr = eh.contact_terrain(st_h["gc"], up["plan_contacts"], ff, leg["foot_pos_abs"], pos[:, 2],
                    st_h["pitch"]); st_h["pitch"] = ctr["root_euler_d_pitch"]
            tick = scen.pack_tick(eul, pos, w, vel, np.c_[np.zeros(n), st_h["pitch"], eul[:, 2]], vd, wd, np.full(n, 0.3))
            sol = eh.solve_ticks(tick, R, leg["foot_pos_abs"], ctr["contacts"])
            st_h["tau"] = eh.joint_torques(act, ctr["contacts"], leg["Jb"], sol["grf"], kin, km, tg, st_h["tau"])
            # ---- device-pointer chain (same inputs uploaded once per tick, everything else stays on the GPU)
            with torch.cuda.stream(st):
                d = {k: T(v) for k, v in dict(q=q, qd=qd, R=R, Rz=Rz, acc=acc, w=w, ff=ff, mm=mm, vd=vd, wd=wd, spd=spd, tg=tg, act=act,
                        eul=eul, z0=np.zeros((n, 3))).items()}
                o = {k: torch.zeros((n, m), dtype=torch.float64, device=dev) for k,
                        m in dict(rel=12, Jb=36, vrel=12, pabs=12, vabs=12, pw=12, vw=12, pos=3, vel=3, trel=12, tabs=12,
                                                                                             tworld=12, cur=12, kin=12, rec=12,
                                                                                                     grf=12).items()}
                ec_d = torch.zeros((n, 4), dtype=torch.uint8, device=dev); pc_d = torch.zeros((n, 4), dtype=torch.uint8,
                        device=dev); ct_d = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
                ta_d = torch.zeros(n, dtype=torch.float64, device=dev); it_d = torch.zeros(n, dtype=torch.int32,
                        device=dev); stt_d = torch.zeros(n, dtype=torch.int32, device=dev)
                L = ed.lib
                assert L.a1mpc_leg_state_batch_device(ed._h, n, ptr(d["q"]), ptr(d["qd"]), ptr(d["R"]), ptr(d["z0"]), ptr(d["z0"]),
                        dp_(fix), dp_(opt), ptr(o["rel"]), ptr(o["Jb"]),
                                                      ptr(o["vrel"]), ptr(o["pabs"]), ptr(o["vabs"]), ptr(o["pw"]), ptr(o["vw"]), sp) == 0
                assert L.a1mpc_ekf_update_batch_device(ed._h, n, 0.0025, 1, ptr(d["mm"]), ptr(d["ff"]), ptr(d["R"]), ptr(d["acc"]),
                        ptr(d["w"]), ptr(o["rel"]), ptr(o["vrel"]),
                                                       ptr(o["pos"]), ptr(o["vel"]), ptr(ec_d), sp) == 0
                assert L.a1mpc_update_plan_batch_device(ed._h, C.byref(gait), n, ptr(d["mm"]), ptr(st_d["gc"]), ptr(d["spd"]),
                        ptr(o["vel"]), ptr(d["Rz"]), ptr(d["R"]), ptr(o["pos"]),
                                                        ptr(d["vd"]), ptr(pc_d), ptr(o["trel"]), ptr(o["tabs"]), ptr(o["tworld"]), sp) == 0
                assert L.a1mpc_swing_legs_batch_device(ed._h, n, 120.0, 0.0025, ptr(d["Rz"]), ptr(o["pabs"]), ptr(st_d["gc"]),
                        ptr(o["trel"]), dp_(kp), dp_(kd), ptr(st_d["start"]),
                                                       ptr(st_d["rl"]), ptr(st_d["tl"]), ptr(o["cur"]), ptr(o["kin"]), sp) == 0
                pz = o["pos"][:, 2].contiguous()
                assert L.a1mpc_contact_terrain_batch_device(ed._h, C.byref(ccfg), n, ptr(st_d["gc"]), ptr(pc_d), ptr(d["ff"]),
                        ptr(o["pabs"]), ptr(pz), ptr(st_d["pitch"]), ptr(ct_d),
                                                            ptr(o["rec"]), ptr(ta_d), sp) == 0
                zc = torch.zeros(n, dtype=torch.float64, device=dev)
                tick_d = torch.cat([d["eul"], o["pos"], d["w"], o["vel"], torch.stack([zc, st_d["pitch"], d["eul"][:, 2]], 1), d["vd"],
                        d["wd"], torch.full((n, 1), 0.3, dtype=torch.float64, device=dev)], 1).contiguous()
                assert L.a1mpc_solve_batch_ticks_device(ed._h, n, ptr(tick_d), ptr(d["R"]), ptr(o["pabs"]), ptr(ct_d), ptr(o["grf"]),
                        None, ptr(it_d), ptr(stt_d), sp) == 0
                assert L.a1mpc_joint_torques_batch_device(ed._h, n, ptr(d["act"]), ptr(ct_d), ptr(o["Jb"]), ptr(o["grf"]), ptr(o["kin"]),
                        dp_(km), ptr(d["tg"]), ptr(st_d["tau"]), sp) == 0
            st.synchronize()
            assert np.array_equal(st_d["tau"].cpu().numpy(), st_h["tau"]), (t, np.abs(st_d["tau"].cpu().numpy() - st_h["tau"]).max())
            assert np.array_equal(o["pos"].cpu().numpy(), pos) and np.array_equal(ct_d.cpu().numpy(),
                    ctr["contacts"]) and np.array_equal(it_d.cpu().numpy(), sol["iters"])


@pytest.mark.parametrize("n,warm,h", [(300, 1, 10), (64, 2, 10), (4096, 1, 10), (300, 2, 12), (64, 1, 6)])   # (h = 12, 6: two of the extended horizons)
def test_control_tick_one_call_matches_the_seven_entry_chain(pkg, scen, n, warm, h):
    """a1mpc_control_tick_device -- leg state, EKF, gait plan, swing legs, contacts / terrain, MPC from tick records and the joint torques in ONE C call,
    N3 inside the MPC kernel's output stage -- against the seven *_device entry points chained by hand on a second handle: every output and every carried state bit for
    bit, four ticks.  n = 300: the fused kernel from the first tick; 64: the latency kernel, update path; 4096: the split pipeline on the first tick (torques by their
    own launch), then the fused kernel in the order of the previous tick's costs (torques in the output stage)."""
    import torch
    rng = np.random.default_rng(2025 + n)
    P = scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS
    cfg = pkg.make_config(P, h, warm_start=warm)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    dp_ = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    E = pkg.engine
    with pkg.Engine(cfg, n, 0) as e1, pkg.Engine(cfg, n, 0) as e7:
        prm = E.TickParams(); e1.lib.a1mpc_default_tick_params(C.byref(prm))
        assert np.allclose(np.array(prm.rho_fix).reshape(4, 5), e1.A1_RHO_FIX) and list(prm.km_foot) == [0.1, 0.1, 0.04]
        kp = np.array(prm.kp_foot); kd = np.array(prm.kd_foot); km = np.array(prm.km_foot); fix = np.array(prm.rho_fix); opt = np.array(prm.rho_opt)
        st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
        init = dict(gait_counter=np.tile([0.0, 120.0, 120.0, 0.0], (n, 1)), root_pos=np.tile([0.0, 0.0, 0.3], (n, 1)))
        state = [{k: T(init.get(k, np.zeros((n, m)))) for k, m in TICK_STATE.items()} for _ in range(2)]
        outs = [{k: torch.zeros((n, m) if m > 1 else (n,), dtype=torch.float64, device=dev) for k, m in TICK_OUT_F64.items()} for _ in range(2)]
        u8 = [{k: torch.zeros((n, 4), dtype=torch.uint8, device=dev) for k in ("estimated_contacts", "plan_contacts", "contacts")} for _ in range(2)]
        i32 = [{k: torch.zeros(n, dtype=torch.int32, device=dev) for k in ("iters", "status")} for _ in range(2)]
        fused_seen = []
        for t in range(4):
            inp = {k: T(v) for k, v in tick_inputs(scen, rng, n).items()}
            # ---- one call
            bf = E.TickBuffers()
            for k in E.TICK_BUFFER_FIELDS:
                src = inp if k in inp else state[0] if k in state[0] else outs[0] if k in outs[0] else u8[0] if k in u8[0] else i32[0]
                setattr(bf, k, src[k].data_ptr())
            e1.control_tick_device(prm, bf, n, stream=st.cuda_stream)
            fused_seen.append(e1.last_control_tick_ms()[1])
            # ---- the chain
            s7, o7, b7, j7, L, H_ = state[1], outs[1], u8[1], i32[1], e7.lib, e7._h
            rcs = [L.a1mpc_leg_state_batch_device(H_, n, ptr(inp["joint_pos"]), ptr(inp["joint_vel"]), ptr(inp["R_world"]), ptr(s7["root_pos"]), ptr(s7["root_lin_vel"]), dp_(fix),
                                                  dp_(opt), ptr(o7["foot_pos_rel"]), ptr(o7["j_foot_blocks"]), ptr(o7["foot_vel_rel"]), ptr(o7["foot_pos_abs"]),
                                                  ptr(o7["foot_vel_abs"]), ptr(o7["foot_pos_world"]), ptr(o7["foot_vel_world"]), sp),
                   L.a1mpc_ekf_update_batch_device(H_, n, prm.control_dt, 1, ptr(inp["movement_mode"]), ptr(inp["foot_force"]), ptr(inp["R_world"]), ptr(inp["imu_acc"]),
                                                   ptr(inp["imu_ang_vel"]), ptr(o7["foot_pos_rel"]), ptr(o7["foot_vel_rel"]), ptr(s7["root_pos"]), ptr(s7["root_lin_vel"]),
                                                   ptr(b7["estimated_contacts"]), sp),
                   L.a1mpc_update_plan_batch_device(H_, C.byref(prm.gait), n, ptr(inp["movement_mode"]), ptr(s7["gait_counter"]), ptr(inp["gait_counter_speed"]),
                                                    ptr(s7["root_lin_vel"]), ptr(inp["R_z"]), ptr(inp["R_world"]), ptr(s7["root_pos"]), ptr(inp["root_lin_vel_d"]),
                                                    ptr(b7["plan_contacts"]), ptr(o7["foot_pos_target_rel"]), ptr(o7["foot_pos_target_abs"]), ptr(o7["foot_pos_target_world"]), sp),
                   L.a1mpc_swing_legs_batch_device(H_, n, prm.gait.counter_per_swing, prm.control_dt, ptr(inp["R_z"]), ptr(o7["foot_pos_abs"]), ptr(s7["gait_counter"]),
                                                   ptr(o7["foot_pos_target_rel"]), dp_(kp), dp_(kd), ptr(s7["foot_pos_start"]), ptr(s7["foot_pos_rel_last_time"]),
                                                   ptr(s7["foot_pos_target_last_time"]), ptr(o7["foot_pos_cur"]), ptr(o7["foot_forces_kin"]), sp)]
            with torch.cuda.stream(st):
                pz = s7["root_pos"][:, 2].contiguous(); pitch = s7["root_euler_d"][:, 1].contiguous()
            rcs.append(L.a1mpc_contact_terrain_batch_device(H_, C.byref(prm.contact), n, ptr(s7["gait_counter"]), ptr(b7["plan_contacts"]), ptr(inp["foot_force"]),
                                                            ptr(o7["foot_pos_abs"]), ptr(pz), ptr(pitch), ptr(b7["contacts"]), ptr(o7["foot_pos_recent_contact"]),
                                                            ptr(o7["terrain_angle"]), sp))
            with torch.cuda.stream(st):
                s7["root_euler_d"][:, 1] = pitch
                tick = torch.cat([inp["root_euler"], s7["root_pos"], inp["root_ang_vel"], s7["root_lin_vel"], s7["root_euler_d"], inp["root_lin_vel_d"], inp["root_ang_vel_d"],
                                  inp["root_pos_d_z"].reshape(n, 1)], 1).contiguous()
            rcs.append(L.a1mpc_solve_batch_ticks_device(H_, n, ptr(tick), ptr(inp["R_world"]), ptr(o7["foot_pos_abs"]), ptr(b7["contacts"]), ptr(o7["grf"]), None,
                                                        ptr(j7["iters"]), ptr(j7["status"]), sp))
            rcs.append(L.a1mpc_joint_torques_batch_device(H_, n, ptr(inp["mpc_active"]), ptr(b7["contacts"]), ptr(o7["j_foot_blocks"]), ptr(o7["grf"]), ptr(o7["foot_forces_kin"]),
                                                          dp_(km), ptr(inp["torques_gravity"]), ptr(s7["joint_torques"]), sp))
            assert not any(rcs), rcs
            st.synchronize()
            for grp in (state, outs, u8, i32):
                for k in grp[0]:
                    a, b = grp[0][k].cpu().numpy(), grp[1][k].cpu().numpy()
                    assert np.array_equal(a, b, equal_nan=True), (t, k, np.abs(a.astype(float) - b.astype(float)).max())
            assert (i32[0]["status"].cpu().numpy() == 1).all() and np.abs(state[0]["joint_torques"].cpu().numpy()).max() > 0.1
        assert fused_seen == ([True] * 4 if n <= 2048 else [False, True, True, True]), fused_seen


# ------------------------------------------------------------------------------------------------------------ round 2


def test_terrain_block_alone(pkg, oracle, scen):
    """a1mpc_terrain_batch (the terrain block of compute_grf with the caller's foot_pos_recent_contact)
    vs the full N2b entry fed the same way."""
    rng = np.random.default_rng(3)
    n = 200
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    base = np.outer([0.2, 0.2, -0.2, -0.2], [1.0, 0.0, 0.3]).reshape(12) + np.outer([1, -1, 1, -1], [0.0, 0.13, 0.0]).reshape(12)
    pitch_a = np.zeros(n); pitch_b = np.zeros(n)
    # all feet in contact: every recent-contact filter updates
    gcs = np.tile([0.0, 0.0, 0.0, 0.0], (n, 1)); plan = np.ones((n, 4), np.uint8)
    with pkg.Engine(cfg, n, 0) as full, pkg.Engine(cfg, n, 0) as only:
        for t in range(130):
            foot = base + rng.normal(0, 0.03, (n, 12)) + np.tile([0.0, 0.0, -0.3], 4); z = np.where(rng.random(n) < 0.9, 0.3,
                    0.05); ff = rng.uniform(0, 80, (n, 4))
            o = full.contact_terrain(gcs, plan, ff, foot, z, pitch_a); pitch_a = o["root_euler_d_pitch"]
            pitch_b, ta = only.terrain(o["foot_pos_recent_contact"], z, pitch_b)
            assert np.array_equal(ta, o["terrain_angle"]) and np.array_equal(pitch_b, pitch_a), t
    assert np.abs(pitch_a).max() > 0.05


@pytest.mark.gpu
def test_ekf_fleet_that_grows_and_is_reset(pkg, oracle, scen):
    """The EKF kernel looks at a robot's initialised-flag where its results leave (round 6: every load of the tick is in flight before anything is computed), so a robot
    whose filter is new in THIS call walks through the arithmetic and must store nothing but its flag.  A fleet of 100 robots that grows to 301 after three ticks (the
    newcomers are initialised in the call that first sees them, the veterans keep filtering), shrinks back to 37, and is reset in the middle: every robot of every
    tick bit for bit against the oracle's device variant, each oracle state started in the tick its robot first appeared."""
    rng = np.random.default_rng(77)
    cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    nmax = 301
    states = [oracle.ekf_state() for _ in range(nmax)]
    sizes = [100, 100, 100, 301, 301, 301, 37, 37, 301, 301, 200, 200]
    with pkg.Engine(cfg, nmax, 0) as eng:
        for t, n in enumerate(sizes):
            if t == 8:
                eng.reset_ekf_state(); states = [oracle.ekf_state() for _ in range(nmax)]
            mm = np.where(rng.random(n) < 0.8, 1, 0).astype(np.uint8)
            yaw = rng.uniform(-3, 3, n); eul = rng.normal(0, 0.05, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9)
            fk = base + rng.normal(0, 0.01, (n, 12)); fv = rng.normal(0, 0.3, (n, 12)); acc = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.3, (n, 3))
            w = rng.normal(0, 0.3, (n, 3)); ff = rng.uniform(0, 160, (n, 4))
            pos, vel, ec = eng.ekf_update(0.0025, mm, ff, R, acc, w, fk, fv)
            for b in range(n):
                p_o, v_o, e_o = oracle.ekf_step(states[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], device=True)
                assert np.array_equal(pos[b], p_o) and np.array_equal(vel[b], v_o) and (ec[b] == e_o).all(), (t, n, b, pos[b] - p_o, vel[b] - v_o)
