"""Pins oracle/a1mpc_oracle.c to the REFERENCE ITSELF (VERDICT r1 task 1).

oracle/_ref/liba1ref*.so = the reference's own sources compiled verbatim from /root/reference (oracle/Makefile `ref`) over
the stand-in headers of oracle/ref_shim/.  Every test drives the reference's classes / functions and the oracle's restatement
with the same numbers.  What is pinned: the reference's SOURCE LOGIC for QP formation (a2-a10), both compute_grf branches
incl. the terrain block (a11, a12), and the caller-side rows N2-N4.  What is not: the rounding of Eigen's kernels (the shim's
products are plain ascending-k sums) and OSQP itself (the OsqpEigen stand-in calls the oracle's restatement).
CPU only.  Skipped when neither /root/reference nor a prebuilt oracle/_ref is present.
"""
import numpy as np
import pytest

import ref as REF

pytestmark = pytest.mark.skipif(not REF.build(), reason="oracle/_ref not built and /root/reference absent")

A1_RHO_FIX = np.array([[0.1805, 0.047, 0.0838, 0.21, 0.21], [0.1805, -0.047, -0.0838, 0.21, 0.21], [-0.1805, 0.047, 0.0838, 0.21, 0.21],
                       [-0.1805, -0.047, -0.0838, 0.21, 0.21]])


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


# ------------------------------------------------------------------------------------------------ formation (a2 - a10)
@pytest.mark.parametrize("h,gen", [(10, "config3_random_flat"), (16, "config4_random_h16"), (20, "config5_divergent")])
def test_formation_equals_reference_ConvexMpc(oracle, scen, h, gen):
    """(P, g, A, l, u) of orc_mpc_form == S/ConvexMpc.cpp compiled verbatim (PLAN_HORIZON = 10 as is, 16 / 20 by the one-macro edit),
    on the random states of BASELINE configs 3 / 4 / 5 and three parameter sets.  Bar: 1e-15 relative to the largest entry (VERDICT)."""
    worst = 0.0
    for ps in ("gazebo", "hardware", "ctrl_default"):
        sc = getattr(scen, gen)(nb=6, param_set=ps) if gen != "config4_random_h16" else scen.config3_random_flat(nb=6, horizon=16, param_set=ps, seed=0xA1 + 4)
        p = sc["params"]
        pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
        for b in range(6):
            P, g, A, l, u, _ = oracle.mpc_form(pr, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b])
            r = REF.convex_mpc_form(h, p["q"], p["r"], sc["x0"][b][:3], p["mass"], p["inertia"], sc["R"][b], sc["foot"][b], sc["contact"][b],
                                    sc["x0"][b], sc["xref"][b], p["dt"])
            assert np.array_equal(A, r["A"]) and np.array_equal(l, r["l"]) and np.array_equal(u, r["u"])
            worst = max(worst, _rel(P, r["P"]), _rel(g, r["g"]))
    assert worst <= 1e-15, worst


def test_formation_per_step_feet_equals_reference(oracle, scen):
    """Per-step B_d (S/ConvexMpc.h:74 B_mat_d_list, S/test/test_mpc.cpp:106-122): feet shifted every horizon step."""
    sc = scen.config3_random_flat(nb=4)
    p = sc["params"]; h = 10
    pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    rng = np.random.default_rng(5)
    for b in range(4):
        vd = rng.uniform(-0.6, 0.6, 3)
        feet = np.stack([sc["foot"][b].reshape(4, 3) - vd * p["dt"] * k for k in range(h)]).reshape(h * 12)
        P, g, A, l, u, _ = oracle.mpc_form(pr, sc["x0"][b], sc["xref"][b], sc["R"][b], feet, sc["contact"][b], foot_stride=12)
        r = REF.convex_mpc_form(h, p["q"], p["r"], sc["x0"][b][:3], p["mass"], p["inertia"], sc["R"][b], feet, sc["contact"][b], sc["x0"][b],
                                sc["xref"][b], p["dt"], foot_stride=12)
        assert _rel(P, r["P"]) <= 1e-15 and _rel(g, r["g"]) <= 1e-15
        P0 = oracle.mpc_form(pr, sc["x0"][b], sc["xref"][b], sc["R"][b], sc["foot"][b], sc["contact"][b])[0]
        assert _rel(P, P0) > 1e-6   # the per-step feet do change the QP


def test_reference_test_mpc_as_written(oracle, scen):
    """S/test/test_mpc.cpp run as written (its main, renamed at compile time): the QP it hands to OSQP equals the oracle's fixture T,
    and the forces it prints are the oracle's default-settings solution of T."""
    text, qp = REF.run_test_mpc()
    sc = scen.scenario_T()
    p = sc["params"]
    pr = oracle.mpc_params(10, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    P, g, A, l, u, csr = oracle.mpc_form(pr, sc["x0"][0], sc["xref"][0], sc["R"][0], sc["foot"][0], sc["contact"][0])
    assert _rel(P, qp["P"]) <= 1e-15 and np.abs(g - qp["q"]).max() <= 1e-18 + 1e-15 * np.abs(g).max()
    assert np.array_equal(A, qp["A"]) and np.array_equal(l, qp["l"]) and np.array_equal(u, qp["u"])
    x, y, info, _ = oracle.osqp_solve(P, g, csr, l, u, oracle.default_settings())
    assert info.iters == qp["iters"] and info.status == qp["status"] == 1
    assert np.abs(x - qp["x"]).max() <= 1e-9
    rows = [[float(v) for v in ln.split()] for ln in text.strip().splitlines()[:3]]
    printed = np.array(rows)                      # 3x4: forces per leg, as std::cout << foot_forces_grf prints them
    assert np.abs(printed - x[:12].reshape(4, 3).T).max() <= 1e-3 * max(1.0, np.abs(x[:12]).max()) * 1e-2   # 6 significant digits
    gold = np.load("tests/golden/T_test_mpc.npz")
    key = "u_default" if "u_default" in gold.files else None
    if key:
        assert np.abs(gold[key][:12] - x[:12]).max() <= 1e-9


# ------------------------------------------------------------------------------------------------ compute_grf (a11, a12)
def _load_state(c, scen, ps, euler, pos, w, v, euler_d, vd, wd, pos_d, foot_abs, contacts, R):
    p = scen.PARAM_SETS[ps]
    c.set("robot_mass", [p["mass"]]); c.set_mat("a1_trunk_inertia", np.asarray(p["inertia"]).reshape(3, 3))
    c.set("q_weights", p["q"]); c.set("r_weights", p["r"])
    c.set("root_euler", euler); c.set("root_pos", pos); c.set("root_ang_vel", w); c.set("root_lin_vel", v)
    c.set("root_euler_d", euler_d); c.set("root_lin_vel_d", vd); c.set("root_ang_vel_d", wd); c.set("root_pos_d", pos_d)
    c.set_mat("root_rot_mat", R); c.set_mat("root_rot_mat_z", scen.rot_zyx(0.0, 0.0, euler[2]))
    c.set("foot_pos_abs", foot_abs); c.set("contacts", contacts)


@pytest.mark.parametrize("h,nt", [(10, 40), (16, 12), (20, 8)])
def test_compute_grf_mpc_branch_warm_sequence(oracle, scen, h, nt):
    """A1RobotControl::compute_grf (stance_leg_control_type = 1) over a warm-started trot sequence vs the oracle chained the
    same way (state packing :452-456, x_ref :470-488, ConvexMpc, persistent warm-started solver :522-538, R'f :555-561), at the reference's
    PLAN_HORIZON = 10 and with the one-macro edit to 16 / 20.  Terrain adaptation off here (it is covered below)."""
    ps = "gazebo"
    sc = scen.config2_trot_sequence(nt, horizon=h)
    p = sc["params"]
    pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    c = REF.Controller(h)
    c.set("stance_leg_control_type", [1]); c.set("use_terrain_adapt", [0])
    carry = oracle.update_carry(h)   # the persistent OsqpEigen solver of the reference: ticks >= 2 take OSQP's update path (S/A1RobotControl.cpp:533-538)
    for t in range(nt):
        x0 = sc["x0"][t]; R = sc["R"][t].reshape(3, 3)
        euler, pos, w, v = x0[0:3], x0[3:6], x0[6:9], x0[9:12]
        vd = np.array([0.3, 0.0, 0.0]); z3 = np.zeros(3)
        _load_state(c, scen, ps, euler, pos, w, v, z3, vd, z3, [0, 0, 0.3], sc["foot"][t], sc["contact"][t], R)
        grf = c.compute_grf(0.0025)
        qp = REF.last_qp(h)
        xref = oracle.mpc_reference(h, p["dt"], euler, pos, R.reshape(9), z3, vd, z3, 0.3)
        assert np.array_equal(xref, sc["xref"][t]) or np.abs(xref - sc["xref"][t]).max() < 1e-15
        assert np.array_equal(c.get("mpc_states_d", 13 * h), xref) and np.array_equal(c.get("mpc_states", 13), x0)
        o = oracle.mpc_solve_update(pr, st, x0, xref, R.reshape(9), sc["foot"][t], sc["contact"][t], carry)
        assert o["info"].iters == qp["iters"] and o["info"].status == qp["status"], t
        # same QP data to 1e-15, same iteration count; the forces then agree to the amplification of that last bit through ~50-100 ADMM iterations
        assert np.abs(o["grf"] - grf).max() <= (1e-9 if h == 10 else 1e-7), (t, np.abs(o["grf"] - grf).max())
    c.close()


def test_compute_grf_balance_branch(oracle, scen):
    """stance_leg_control_type = 0 (S/A1RobotControl.cpp:377-444, ctor :11-48): PD wrench, P / q / bounds, cold OSQP, R'f -- 40 random states."""
    rng = np.random.default_rng(3)
    qp_par = oracle.default_qp_params(); st = oracle.default_settings(warm_start=0)
    c = REF.Controller()
    c.set("stance_leg_control_type", [0])
    kp_lin, kd_lin, kp_ang, kd_ang = [1000.0] * 3, [200.0, 70.0, 120.0], [650.0, 35.0, 1.0], [4.5, 4.5, 30.0]   # S/A1CtrlStates.h:117-120
    for k in range(40):
        euler = np.array([rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15), rng.uniform(-np.pi, np.pi)])
        R = scen.rot_zyx(*euler); Rz = scen.rot_zyx(0.0, 0.0, euler[2])
        pos = rng.normal(0, 0.5, 3); pos[2] = rng.uniform(0.2, 0.32); pos_d = pos + rng.normal(0, 0.02, 3)
        w = rng.normal(0, 0.3, 3); v = rng.normal(0, 0.3, 3); vd = rng.uniform(-0.5, 0.5, 3); wd = rng.uniform(-0.5, 0.5, 3)
        euler_d = euler + rng.normal(0, 0.05, 3)
        if k == 7:
            euler_d[2] = euler[2] + 5.0   # the yaw wrap branch (:328-332)
        nominal = np.array(scen.PARAM_SETS["gazebo"]["foot"]); foot = (R @ (nominal + rng.uniform(-0.03, 0.03, (4, 3))).T).T.reshape(12)
        contacts = (rng.random(4) < 0.7).astype(np.uint8) if k % 3 else np.ones(4, np.uint8)
        _load_state(c, scen, "gazebo", euler, pos, w, v, euler_d, vd, wd, pos_d, foot, contacts, R)
        grf = c.compute_grf(0.0025)
        qp = REF.last_qp()
        acc = oracle.balance_root_acc(kp_lin, kd_lin, kp_ang, kd_ang, pos_d, pos, vd, v, euler_d, euler, wd, w, R.reshape(9), 12.0)
        P, g, A, l, u, _ = oracle.balance_form(qp_par, acc, Rz.reshape(9), foot, contacts)
        assert _rel(P, qp["P"]) <= 1e-15 and _rel(g, qp["q"]) <= 1e-14, (k, _rel(P, qp["P"]), _rel(g, qp["q"]))
        assert np.array_equal(A, qp["A"]) and np.array_equal(l, qp["l"]) and np.array_equal(u, qp["u"])
        o = oracle.balance_solve(qp_par, st, acc, R.reshape(9), Rz.reshape(9), foot, contacts)
        assert o["info"].iters == qp["iters"] and o["info"].status == qp["status"]
        assert np.abs(o["grf"] - grf).max() <= 1e-7, (k, np.abs(o["grf"] - grf).max())
    c.close()


# ------------------------------------------------------------------------------------------------ caller-side rows
def test_moving_window_filter_is_the_reference_header(oracle):
    """S/utils/filter.hpp used as is vs the oracle's filter (inside orc_contact_terrain_step): the terrain-angle filter (window 100)
    and the contact filters (window 60) are exercised through the controller below; here the raw class on 1e5 samples, bit for bit
    against an independent restatement of the Neumaier update."""
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, 100000) * np.exp(rng.uniform(-8, 8, 100000))
    for win in (60, 100, 7):
        got = REF.filter_run(win, x)
        s = 0.0; c = 0.0; out = np.zeros_like(x); q = []
        for k, v in enumerate(x):
            def upd(val):
                nonlocal s, c
                ns = s + val
                c += ((s - ns) + val) if abs(s) >= abs(val) else ((val - ns) + s)
                s = ns
            if len(q) >= win:
                upd(-q.pop(0))
            upd(v); q.append(v)
            out[k] = (s + c) / float(win)
        assert np.array_equal(got, out), win


def test_caller_side_tick_chain_equals_reference(oracle, scen):
    """update_plan -> generate_swing_legs_ctrl -> compute_grf (terrain block + MPC) -> compute_joint_torques of the reference, 150 ticks,
    vs the oracle functions chained the same way (orc_update_plan, orc_swing_legs, orc_contact_terrain_step, orc_mpc_solve, orc_joint_torques).
    Element-wise rows are compared bit for bit; the plane fit goes through the shim's Jacobi SVD here and a Jacobi eigen-solver in
    the oracle (1e-9 on the angle); forces within 1e-6 N as long as the iteration counts agree (asserted)."""
    rng = np.random.default_rng(17)
    ps = "gazebo"; p = dict(scen.PARAM_SETS[ps], **scen.MPC_CONSTANTS)
    pr = oracle.mpc_params(10, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    c = REF.Controller()
    c.set("stance_leg_control_type", [1]); c.set("use_terrain_adapt", [1]); c.set("movement_mode", [1])
    default_foot = np.array([[0.17, 0.15, -0.35], [0.17, -0.15, -0.35], [-0.17, 0.15, -0.35], [-0.17, -0.15, -0.35]])
    c.set("default_foot_pos", default_foot.reshape(12))
    gp = oracle.gait_params(default_foot.reshape(12))
    cstate = oracle.contact_state()
    gc = np.array([0.0, 120.0, 120.0, 0.0]); spd = np.array([2.0, 2.0, 2.0, 2.0])
    st3 = [np.zeros(12) for _ in range(3)]
    wx = np.zeros(120); wy = np.zeros(200); rho = 0.0
    pitch_d = 0.0; tau_prev = np.zeros(12); grf_prev = np.zeros(12)
    km = np.array([0.1, 0.1, 0.1]); tg = np.array([0.8, 0, 0, -0.8, 0, 0, 0.8, 0, 0, -0.8, 0, 0])
    slope = 0.25
    carry_mpc = oracle.update_carry(10)
    for t in range(150):
        euler = np.array([rng.normal(0, 0.02), 0.2 + rng.normal(0, 0.02), 0.3 + 0.001 * t])
        R = scen.rot_zyx(*euler); Rz = scen.rot_zyx(0.0, 0.0, euler[2])
        pos = np.array([0.002 * t, 0.0, 0.3 + rng.normal(0, 0.005)]); w = rng.normal(0, 0.1, 3); v = np.array([0.3, 0, 0]) + rng.normal(0, 0.05, 3)
        vd = np.array([0.3, 0.0, 0.0]); wd = np.array([0.0, 0.0, 0.1])
        foot_abs = (R @ (default_foot + rng.normal(0, 0.02, (4, 3))).T).T
        foot_abs[:, 2] += slope * foot_abs[:, 0]           # feet on a slope => non-zero terrain angle
        foot_abs = foot_abs.reshape(12)
        ff = rng.uniform(0, 80, 4)
        Jb = rng.normal(0, 0.2, (4, 9)); Jb[:, [0, 4, 8]] += 0.3
        # ---- reference
        c.set("gait_counter", gc); c.set("gait_counter_speed", spd)
        _load_state(c, scen, ps, euler, pos, w, v, [0.0, pitch_d, 0.0], vd, wd, [0, 0, 0.3], foot_abs, c.get("contacts", 4), R)
        c.set("foot_force", ff)
        J = np.zeros((12, 12))
        for i in range(4):
            J[3 * i:3 * i + 3, 3 * i:3 * i + 3] = Jb[i].reshape(3, 3).T   # Jb blocks are column-major
        c.set_mat("j_foot", J); c.set("km_foot", km); c.set("torques_gravity", tg)
        c.update_plan(0.0025)
        c.generate_swing_legs_ctrl(0.0025)
        grf_r = c.compute_grf(0.0025)
        qp = REF.last_qp()
        c.set("foot_forces_grf", grf_r)
        c.compute_joint_torques()
        # ---- oracle
        gc_o, plan, rel, ab, wo = oracle.update_plan(gp, 1, gc, spd, v, Rz.reshape(9), R.reshape(9), pos, vd)
        assert np.array_equal(gc_o, c.get("gait_counter", 4)) and np.array_equal(plan, c.get("plan_contacts", 4).astype(np.uint8)), t
        assert np.array_equal(rel, c.get("foot_pos_target_rel", 12)) and np.array_equal(ab, c.get("foot_pos_target_abs", 12)) and np.array_equal(wo, c.get("foot_pos_target_world", 12)), t
        cur, kin = oracle.swing_legs(Rz.reshape(9), foot_abs, gc_o, rel, st3[0], st3[1], st3[2])
        assert np.array_equal(cur, c.get("foot_pos_cur", 12)) and np.array_equal(st3[0], c.get("foot_pos_start", 12)), t
        assert np.abs(st3[2] - c.get("foot_pos_target_last_time", 12)).max() <= 1e-15 and np.abs(kin - c.get("foot_forces_kin", 12)).max() <= 1e-9, t
        st3[2] = c.get("foot_pos_target_last_time", 12).copy(); kin = c.get("foot_forces_kin", 12).copy()
        ct, rec, ang, pitch_o = oracle.contact_terrain_step(cstate, gc_o, plan, ff, foot_abs, pos[2], pitch_d)
        assert np.array_equal(ct, c.get("contacts", 4).astype(np.uint8)) and np.array_equal(rec, c.get("foot_pos_recent_contact", 12)), t
        assert abs(ang - c.get("terrain_pitch_angle", 1)[0]) <= 1e-9 and abs(pitch_o - c.get("root_euler_d", 3)[1]) <= 1e-9, (t, ang, c.get("terrain_pitch_angle", 1))
        pitch_d = c.get("root_euler_d", 3)[1]
        xref = oracle.mpc_reference(10, p["dt"], euler, pos, R.reshape(9), [0.0, pitch_d, 0.0], vd, wd, 0.3)
        assert np.array_equal(xref, c.get("mpc_states_d", 130)), t
        x0 = np.concatenate([euler, pos, w, v, [-9.8]])
        o = oracle.mpc_solve_update(pr, st, x0, xref, R.reshape(9), foot_abs, ct, carry_mpc)   # (the reference's persistent solver: OSQP's update path from tick 2 on)
        assert o["info"].iters == qp["iters"] and o["info"].status == qp["status"], t
        assert np.abs(o["grf"] - grf_r).max() <= 1e-6, (t, np.abs(o["grf"] - grf_r).max())
        tau = oracle.joint_torques(1 if t >= 9 else 0, ct, Jb.reshape(36), grf_r, kin, km, tg, tau_prev)
        assert np.array_equal(tau, c.get("joint_torques", 12)), (t, tau - c.get("joint_torques", 12))
        tau_prev = tau; gc = gc_o
    assert abs(pitch_d) > 0.05      # the terrain block did something
    c.close()


def test_leg_kinematics_equals_reference(oracle):
    """A1Kinematics::fk / jac (MATLAB-generated, S/legKinematics/A1Kinematics.cpp:39-131) vs the oracle's restatement from the leg model."""
    rng = np.random.default_rng(2)
    worst = 0.0
    for k in range(300):
        q = rng.uniform(-1.5, 1.5, 12); qd = rng.normal(0, 3, 12); opt = rng.normal(0, 0.01, (4, 3))
        R = np.eye(3).reshape(9)
        o = oracle.leg_state(q, qd, R, np.zeros(3), np.zeros(3), rho_opt=opt)
        for leg in range(4):
            p = REF.leg_fk(q[3 * leg:3 * leg + 3], opt[leg], A1_RHO_FIX[leg])
            J = REF.leg_jac(q[3 * leg:3 * leg + 3], opt[leg], A1_RHO_FIX[leg])
            worst = max(worst, np.abs(p - o["foot_pos_rel"][3 * leg:3 * leg + 3]).max(), np.abs(J - o["Jb"][9 * leg:9 * leg + 9]).max())
    assert worst <= 2e-15, worst


def test_bezier_and_utils_equal_reference(oracle, scen):
    """BezierUtils::get_foot_pos_curve (S/utils/Utils.cpp:64-106) is exercised through generate_swing_legs_ctrl above; here skew,
    quat_to_euler, pseudo_inverse and cal_dihedral_angle against numpy."""
    rng = np.random.default_rng(9)
    for _ in range(50):
        v = rng.normal(0, 1, 3)
        assert np.array_equal(REF.skew(v), np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]]))
        e = np.array([rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3)])
        R = scen.rot_zyx(*e)
        w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        if w > 0.1:
            x, y, z = (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)
            assert np.abs(REF.quat_to_euler(w, x, y, z) - e).max() <= 1e-12
        M = rng.normal(0, 1, (3, 3)); M = M @ M.T
        assert np.abs(REF.pseudo_inverse(M) - np.linalg.pinv(M)).max() <= 1e-9 * np.abs(np.linalg.pinv(M)).max()
        a, b = rng.normal(0, 1, 3), rng.normal(0, 1, 3)
        assert abs(REF.dihedral_angle(a, b) - np.arccos(abs(a @ b) / np.linalg.norm(a) / np.linalg.norm(b))) <= 1e-14
    M = np.outer([1.0, 2.0, 3.0], [1.0, 2.0, 3.0])   # rank 1: the start-up case of the plane fit
    assert np.abs(REF.pseudo_inverse(M) - np.linalg.pinv(M)).max() <= 1e-12
    t = np.float32(0.37)
    out = REF.bezier_foot_curve(float(t), [0.1, 0.2, -0.3], [0.2, 0.1, -0.3])
    tt = float(t)
    bz = lambda P: sum(cf * tt ** i * (1 - tt) ** (4 - i) * P[i] for i, cf in enumerate([1, 4, 6, 4, 1]))
    assert abs(out[0] - bz([0.1, 0.1, 0.2, 0.2, 0.2])) <= 1e-15 and abs(out[2] - bz([-0.3, -0.3, -0.3 + float(np.float32(0.4)), -0.3, -0.3])) <= 1e-15


def test_ekf_equals_reference(oracle, scen):
    """A1BasicEKF (S/A1BasicEKF.cpp) compiled verbatim, 120 ticks, vs orc_ekf_step.  The two fullPivHouseholderQr().solve() calls are a
    partial-pivot elimination in the shim and a Gauss-Jordan in the oracle: agreement to solver accuracy (1e-9), not bit for bit."""
    rng = np.random.default_rng(51)
    c = REF.Controller(); c.ekf_new(True)
    state = oracle.ekf_state()
    base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
    worst = 0.0
    for t in range(120):
        mm = 1 if (t > 3 and rng.random() < 0.8) else 0
        yaw = rng.uniform(-3, 3); e = rng.normal(0, 0.05, 2); R = scen.rot_zyx(e[0], e[1], yaw)
        fk = base + rng.normal(0, 0.01, 12); fv = rng.normal(0, 0.3, 12); acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.3, 3)
        w = rng.normal(0, 0.3, 3); ff = rng.uniform(0, 160, 4)
        c.set("movement_mode", [mm]); c.set("foot_force", ff); c.set_mat("root_rot_mat", R); c.set("imu_acc", acc); c.set("imu_ang_vel", w)
        c.set("foot_pos_rel", fk); c.set("foot_vel_rel", fv)
        if t == 0:
            c.ekf_init_state()      # S/GazeboA1ROS.cpp:163-165: first tick initialises, later ticks update
        else:
            c.ekf_update(0.0025)
        p_o, v_o, e_o = oracle.ekf_step(state, 0.0025, mm, ff, R.reshape(9), acc, w, fk, fv)
        if t > 0:
            worst = max(worst, np.abs(p_o - c.get("estimated_root_pos", 3)).max(), np.abs(v_o - c.get("estimated_root_vel", 3)).max())
            assert np.array_equal(e_o, c.get("estimated_contacts", 4).astype(np.uint8)), t
    assert worst <= 1e-9, worst
    c.close()


def test_update_path_reinitialises_when_the_hessian_pattern_changes(oracle, scen):
    """VERDICT r2 (missing 1) / SURVEY 8(c): the reference's Hessian is dense.sparseView() (S/ConvexMpc.cpp:211), so exact zeros appearing or vanishing change its
    sparsity pattern, and osqp-eigen's updateHessianMatrix then does NOT take osqp_update_P: it reads the workspace iterates, clears and re-initialises the solver
    (rho back to settings.rho, fresh scaling) and warm-starts it with them (S/A1RobotControl.cpp:533-538).  Fixture T (S/test/test_mpc.cpp:18-60: level, yaw = 0,
    symmetric feet) is such a state.  The reference's own compute_grf is walked from T into a pitched state and back; the stand-in's updateHessianMatrix compares the
    triplet patterns like osqp-eigen, the oracle's orc_mpc_solve_update compares the patterns of the dense P it forms itself: same re-initialisation ticks, same
    iteration counts, same forces."""
    h = 10
    T = scen.scenario_T()
    p = T["params"]
    pr = oracle.mpc_params(h, p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = oracle.default_settings(warm_start=1)
    c = REF.Controller(h)
    c.set("stance_leg_control_type", [1]); c.set("use_terrain_adapt", [0])
    carry = oracle.update_carry(h)
    # With fixture T's weights (zeros on x / y position and roll / pitch rate) the level state leaves 608 exact zeros in the upper triangle of P; any roll or pitch
    # rotates the inertia and fills them (a yaw alone does not: T's attitude weights are isotropic).  (The controller's own weight sets -- Gazebo / hardware /
    # Isaac -- give a fully dense P in every state: this branch is the interface's, fixture T's, not the walking robot's.)
    yaws = [0.0, 0.0, 0.02, 0.03, 0.03, 0.0, 0.0, 0.05, 0.0]      # pitch; the pattern changes entering tick 2 (zeros vanish), 5 (they come back), 7, 8
    nominal = np.array([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35]).reshape(4, 3)
    seen = []; nnzs = []
    for t, yaw in enumerate(yaws):
        euler = np.array([0.0, yaw, 0.0]); pos = np.array([0.0, 0.0, 0.15 + 0.001 * t]); z3 = np.zeros(3)
        R = scen.rot_zyx(0.0, yaw, 0.0)
        foot = (R @ nominal.T).T.reshape(12) if yaw != 0.0 else nominal.reshape(12)
        contacts = np.array([1, 0, 1, 0], np.uint8)
        c.set("robot_mass", [p["mass"]]); c.set_mat("a1_trunk_inertia", np.asarray(p["inertia"]).reshape(3, 3))
        c.set("q_weights", p["q"]); c.set("r_weights", p["r"])
        c.set("root_euler", euler); c.set("root_pos", pos); c.set("root_ang_vel", z3); c.set("root_lin_vel", z3)
        c.set("root_euler_d", z3); c.set("root_lin_vel_d", z3); c.set("root_ang_vel_d", z3); c.set("root_pos_d", [0, 0, 0.15])
        c.set_mat("root_rot_mat", R); c.set_mat("root_rot_mat_z", np.eye(3)); c.set("foot_pos_abs", foot); c.set("contacts", contacts)
        grf = c.compute_grf(0.0025)
        qp = REF.last_qp(h)
        x0 = np.concatenate([euler, pos, z3, z3, [-9.8]])
        xref = oracle.mpc_reference(h, p["dt"], euler, pos, R.reshape(9), z3, z3, z3, 0.15)
        assert np.array_equal(c.get("mpc_states", 13), x0) and np.array_equal(c.get("mpc_states_d", 13 * h), xref)
        o = oracle.mpc_solve_update(pr, st, x0, xref, R.reshape(9), foot, contacts, carry)
        seen.append(qp["reinit"])
        assert o["info"].reinit == qp["reinit"], (t, o["info"].reinit, qp["reinit"])
        assert o["info"].iters == qp["iters"] and o["info"].status == qp["status"], (t, o["info"].iters, qp["iters"])
        assert np.abs(o["grf"] - grf).max() <= 1e-9, (t, np.abs(o["grf"] - grf).max())
        nnzs.append(int((np.triu(qp["P"]) != 0).sum()))
    c.close()
    assert seen == [0, 0, 1, 0, 0, 1, 0, 1, 1], seen
    level = {n_ for n_, y_ in zip(nnzs, yaws) if y_ == 0.0}; yawed = {n_ for n_, y_ in zip(nnzs, yaws) if y_ != 0.0}
    assert len(level) == 1 and len(yawed) == 1 and max(level) < min(yawed) <= 120 * 121 // 2, (nnzs,)   # the level state has exact zeros in P, the pitched one none
