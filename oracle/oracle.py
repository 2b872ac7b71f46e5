"""ctypes wrapper around oracle/liba1mpc_oracle.so  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of a1mpc_oracle.c).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liba1mpc_oracle.so")

NS, NU, NC = 13, 12, 20

STATUS = {1: "solved", 2: "solved_inaccurate", -2: "max_iter", -3: "primal_infeasible", -4: "dual_infeasible",
          -7: "non_cvx", -10: "unsolved"}


class Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("eps_abs", C.c_double),
                ("eps_rel", C.c_double), ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("adaptive_rho_tolerance", C.c_double), ("max_iter", C.c_int32), ("scaling", C.c_int32),
                ("check_termination", C.c_int32), ("adaptive_rho", C.c_int32), ("adaptive_rho_interval", C.c_int32),
                ("warm_start", C.c_int32), ("linsys", C.c_int32), ("reserved_", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("iters", C.c_int32), ("status", C.c_int32), ("rho_updates", C.c_int32), ("nfact", C.c_int32),
                ("pri_res", C.c_double), ("dua_res", C.c_double), ("rho_final", C.c_double), ("reinit", C.c_int32), ("pad_", C.c_int32)]


class MpcParams(C.Structure):
    _fields_ = [("horizon", C.c_int32), ("dt", C.c_double), ("mu", C.c_double), ("fz_min", C.c_double),
                ("fz_max", C.c_double), ("q", C.c_double * NS), ("r", C.c_double * NU), ("mass", C.c_double),
                ("inertia", C.c_double * 9)]


class QpParams(C.Structure):
    _fields_ = [("Qw", C.c_double * 6), ("R", C.c_double), ("mu", C.c_double), ("F_min", C.c_double),
                ("F_max", C.c_double)]


def build(force=False):
    """Compile the C restatement (gcc, a few seconds)."""
    src = os.path.join(_HERE, "a1mpc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liba1mpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a, t=C.c_double):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


def default_settings(**over):
    s = Settings()
    lib().orc_default_settings(C.byref(s))
    for k, v in over.items():
        setattr(s, k, v)
    return s


def exact_settings(**over):
    """'exact' mode: same algorithm driven to a tight tolerance (the truth the GPU and
    the default-tolerance oracle are both measured against)."""
    kw = dict(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000)
    kw.update(over)
    return default_settings(**kw)


def mpc_params(horizon, dt, mu, fz_min, fz_max, q, r, mass, inertia):
    p = MpcParams()
    p.horizon, p.dt, p.mu, p.fz_min, p.fz_max, p.mass = int(horizon), dt, mu, fz_min, fz_max, mass
    p.q[:] = list(np.asarray(q, dtype=float))
    p.r[:] = list(np.asarray(r, dtype=float))
    p.inertia[:] = list(np.asarray(inertia, dtype=float).reshape(9))
    return p


def default_qp_params():
    p = QpParams()
    lib().orc_default_qp_params(C.byref(p))
    return p


def mpc_form(pr, x0, xref, Rw, foot, contact, yaw=None, foot_stride=0, contact_stride=0):
    """Dense QP data of one problem: P (n,n), g, A (m,n dense, from CSR), l, u."""
    h = pr.horizon
    n, m = NU * h, NC * h
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    xref = np.ascontiguousarray(xref, dtype=np.float64)
    Rw = np.ascontiguousarray(Rw, dtype=np.float64)
    foot = np.ascontiguousarray(foot, dtype=np.float64)
    contact = np.ascontiguousarray(contact, dtype=np.uint8)
    P = np.zeros((n, n)); g = np.zeros(n); l = np.zeros(m); u = np.zeros(m)
    rp = np.zeros(m + 1, dtype=np.int32); ci = np.zeros(36 * h, dtype=np.int32); av = np.zeros(36 * h)
    lib().orc_mpc_form(C.byref(pr), _p(x0), _p(xref), C.c_double(float(x0[2] if yaw is None else yaw)), _p(Rw), _p(foot),
                       C.c_int(foot_stride), _p(contact, C.c_uint8), C.c_int(contact_stride), _p(P), _p(g),
                       _p(rp, C.c_int32), _p(ci, C.c_int32), _p(av), _p(l), _p(u))
    A = np.zeros((m, n))
    for i in range(m):
        for k in range(rp[i], rp[i + 1]):
            A[i, ci[k]] = av[k]
    return P, g, A, l, u, (rp, ci, av)


def osqp_solve(P, g, csr, l, u, st, x=None, y=None, rho=None):
    rp, ci, av = csr
    n, m = len(g), len(l)
    P = np.ascontiguousarray(P, dtype=np.float64)
    x = np.zeros(n) if x is None else np.array(x, dtype=np.float64)
    y = np.zeros(m) if y is None else np.array(y, dtype=np.float64)
    info = Info()
    rho_io = C.c_double(0.0 if rho is None else rho)
    lib().orc_osqp_solve(C.c_int(n), C.c_int(m), _p(P), _p(np.ascontiguousarray(g)), _p(rp, C.c_int32), _p(ci, C.c_int32),
                         _p(av), _p(np.ascontiguousarray(l)), _p(np.ascontiguousarray(u)), C.byref(st), _p(x), _p(y),
                         C.byref(rho_io), C.byref(info))
    return x, y, info, rho_io.value


def last_z(m):
    """unscaled z of this thread's most recent single solve (mpc_solve / mpc_solve_update / osqp_solve), reference row order"""
    z = np.zeros(m)
    assert lib().orc_last_z(C.c_int(m), _p(z)) == m
    return z


def check_termination(P, g, csr, x, z, y, eps_abs=1e-3, eps_rel=1e-3):
    """OSQP's termination test on a GIVEN unscaled iterate (x, z, y) of the QP (P, g, A = csr): dict(ok, pri_res, pri_tol, dua_res, dua_tol).
    For the ticks where engine and oracle stop at different iterations: is the engine's point one OSQP itself would have stopped at?"""
    rp, ci, av = csr
    n, m = len(g), len(rp) - 1
    out = np.zeros(4)
    ok = lib().orc_check_termination(C.c_int(n), C.c_int(m), _p(np.ascontiguousarray(P, dtype=np.float64)), _p(np.ascontiguousarray(g, dtype=np.float64)),
                                     _p(rp, C.c_int32), _p(ci, C.c_int32), _p(av), _p(np.ascontiguousarray(x, dtype=np.float64)),
                                     _p(np.ascontiguousarray(z, dtype=np.float64)), _p(np.ascontiguousarray(y, dtype=np.float64)),
                                     C.c_double(eps_abs), C.c_double(eps_rel), _p(out))
    return dict(ok=bool(ok == 1), pri_res=float(out[0]), pri_tol=float(out[1]), dua_res=float(out[2]), dua_tol=float(out[3]))


def mpc_solve_batch(pr, st, x0, xref, Rw, foot, contact, nthreads=0, want_u=False):
    """Batch of independent ticks (S/A1RobotControl.cpp:446-562 each).  Arrays are (nb, ...)."""
    h = pr.horizon
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, NS)
    nb = x0.shape[0]
    xref = np.ascontiguousarray(xref, dtype=np.float64).reshape(nb, NS * h)
    Rw = np.ascontiguousarray(Rw, dtype=np.float64).reshape(nb, 9)
    foot = np.ascontiguousarray(foot, dtype=np.float64).reshape(nb, 12)
    contact = np.ascontiguousarray(contact, dtype=np.uint8).reshape(nb, 4)
    grf = np.zeros((nb, 12)); u = np.zeros((nb, NU * h)) if want_u else None
    iters = np.zeros(nb, dtype=np.int32); status = np.zeros(nb, dtype=np.int32); nfact = np.zeros(nb, dtype=np.int32)
    lib().orc_mpc_solve_batch(C.byref(pr), C.byref(st), C.c_int(nb), _p(x0), _p(xref), _p(Rw), _p(foot), _p(contact, C.c_uint8),
                              _p(grf), _p(u), _p(iters, C.c_int32), _p(status, C.c_int32), _p(nfact, C.c_int32),
                              C.c_int(nthreads))
    return dict(grf=grf, u=u, iters=iters, status=status, nfact=nfact)


def mpc_solve(pr, st, x0, xref, Rw, foot, contact, warm_x=None, warm_y=None, warm_rho=None, foot_stride=0,
              contact_stride=0):
    h = pr.horizon
    n, m = NU * h, NC * h
    grf = np.zeros(12); u = np.zeros(n)
    info = Info()
    wx = None if warm_x is None else np.array(warm_x, dtype=np.float64)
    wy = None if warm_y is None else np.array(warm_y, dtype=np.float64)
    rho = C.c_double(0.0 if warm_rho is None else warm_rho)
    lib().orc_mpc_solve(C.byref(pr), C.byref(st), _p(np.ascontiguousarray(x0, dtype=np.float64)),
                        _p(np.ascontiguousarray(xref, dtype=np.float64)), _p(np.ascontiguousarray(Rw, dtype=np.float64)),
                        _p(np.ascontiguousarray(foot, dtype=np.float64)), C.c_int(foot_stride),
                        _p(np.ascontiguousarray(contact, dtype=np.uint8), C.c_uint8), C.c_int(contact_stride), _p(grf), _p(u),
                        _p(wx), _p(wy), C.byref(rho), C.byref(info))
    return dict(grf=grf, u=u, info=info, warm_x=wx, warm_y=wy, rho=rho.value)


def update_carry(horizon):
    """zeroed workspace carry of mpc_solve_update for one robot (2 + 2n + 4m doubles + 2 for the sparsity pattern of the previous tick's P)"""
    return np.zeros(4 + 2 * NU * horizon + 4 * NC * horizon)


def carry_from_workspace(h, x, y, z, rho, D, E, c, P_prev, g_prev, l_prev, u_prev):
    """the workspace carry of mpc_solve_update (osqp_solve_impl's layout: valid, rho, x_s, z_s, y_s, q_prev, l_prev, u_prev, pattern signature of the previous P)
    built from an UNSCALED state (x, y, z, rho) and the equilibration (D, E, c) it was reached under -- x_s = x / D, z_s = E z, y_s = c y / E -- plus the previous
    tick's QP data.  How the tests start this oracle from the engine's workspace (a1mpc_get_warm_start / _workspace_z / _workspace_scaling)."""
    n, m = NU * h, NC * h
    carry = update_carry(h)
    carry[0] = 1.0; carry[1] = rho
    o = 2
    carry[o:o + n] = np.asarray(x) / np.asarray(D); o += n
    carry[o:o + m] = np.asarray(E) * np.asarray(z); o += m
    carry[o:o + m] = c * np.asarray(y) / np.asarray(E); o += m
    carry[o:o + n] = g_prev; o += n
    carry[o:o + m] = l_prev; o += m
    carry[o:o + m] = u_prev; o += m
    sig = np.zeros(2)
    lib().orc_pattern_signature(C.c_int(n), _p(np.ascontiguousarray(P_prev, dtype=np.float64)), _p(sig))
    carry[o:o + 2] = sig
    return carry


def mpc_solve_update(pr, st, x0, xref, Rw, foot, contact, carry, foot_stride=0, contact_stride=0, yaw=None):
    """one tick on the reference's UPDATE PATH (updateHessianMatrix / updateGradient / update*Bound on a persistent OSQP workspace,
    S/A1RobotControl.cpp:533-538; see osqp_solve_impl in a1mpc_oracle.c).  `carry` is updated in place.  foot_stride / contact_stride / yaw: the general case of the
    reference's interface (per-step feet, a contact schedule, A_c from another yaw), as in mpc_solve."""
    h = pr.horizon
    grf = np.zeros(12); u = np.zeros(NU * h); info = Info()
    assert carry.dtype == np.float64 and carry.size == 4 + 2 * NU * h + 4 * NC * h and carry.flags.c_contiguous
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    lib().orc_mpc_solve_update_strided(C.byref(pr), C.byref(st), _p(x0), _p(np.ascontiguousarray(xref, dtype=np.float64)), C.c_double(float(x0[2] if yaw is None else yaw)),
                                       _p(np.ascontiguousarray(Rw, dtype=np.float64)), _p(np.ascontiguousarray(foot, dtype=np.float64)), C.c_int(int(foot_stride)),
                                       _p(np.ascontiguousarray(contact, dtype=np.uint8), C.c_uint8), C.c_int(int(contact_stride)), _p(grf), _p(u), _p(carry), C.byref(info))
    return dict(grf=grf, u=u, info=info)


def mpc_reference(h, dt, euler, pos, Rw, euler_d, lin_vel_d_body, ang_vel_d, pos_z_d):
    xref = np.zeros(NS * h)
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    lib().orc_mpc_reference(C.c_int(h), C.c_double(dt), a(euler), a(pos), a(Rw), a(euler_d), a(lin_vel_d_body), a(ang_vel_d),
                            C.c_double(pos_z_d), _p(xref))
    return xref


def balance_form(qp, root_acc, Rz, foot, contact):
    P = np.zeros((12, 12)); g = np.zeros(12); l = np.zeros(20); u = np.zeros(20)
    rp = np.zeros(21, dtype=np.int32); ci = np.zeros(36, dtype=np.int32); av = np.zeros(36)
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    lib().orc_balance_form(C.byref(qp), a(root_acc), a(Rz), a(foot), _p(np.ascontiguousarray(contact, dtype=np.uint8), C.c_uint8),
                           _p(P), _p(g), _p(rp, C.c_int32), _p(ci, C.c_int32), _p(av), _p(l), _p(u))
    A = np.zeros((20, 12))
    for i in range(20):
        for k in range(rp[i], rp[i + 1]):
            A[i, ci[k]] = av[k]
    return P, g, A, l, u, (rp, ci, av)


def balance_solve(qp, st, root_acc, Rw, Rz, foot, contact):
    grf = np.zeros(12); f = np.zeros(12); info = Info()
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    lib().orc_balance_solve(C.byref(qp), C.byref(st), a(root_acc), a(Rw), a(Rz), a(foot),
                            _p(np.ascontiguousarray(contact, dtype=np.uint8), C.c_uint8), _p(grf), _p(f), C.byref(info))
    return dict(grf=grf, f_world=f, info=info)


def balance_root_acc(kp_lin, kd_lin, kp_ang, kd_ang, pos_d, pos, lin_vel_d_body, lin_vel_world, euler_d, euler,
                     ang_vel_d_body, ang_vel_world, Rw, mass):
    out = np.zeros(6)
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    lib().orc_balance_root_acc(a(kp_lin), a(kd_lin), a(kp_ang), a(kd_ang), a(pos_d), a(pos), a(lin_vel_d_body), a(lin_vel_world),
                               a(euler_d), a(euler), a(ang_vel_d_body), a(ang_vel_world), a(Rw), C.c_double(mass), _p(out))
    return out


class GaitParams(C.Structure):
    _fields_ = [("counter_per_gait", C.c_double), ("counter_per_swing", C.c_double), ("control_dt", C.c_double),
                ("foot_delta_x_limit", C.c_double), ("foot_delta_y_limit", C.c_double), ("default_foot_pos", C.c_double * 12),
                ("gait_counter_reset", C.c_double * 4)]


def gait_params(default_foot_pos, counter_per_gait=240.0, counter_per_swing=120.0, control_dt=0.0025, dx=0.1, dy=0.1,
                reset=(0.0, 120.0, 120.0, 0.0)):
    p = GaitParams()
    p.counter_per_gait, p.counter_per_swing, p.control_dt, p.foot_delta_x_limit, p.foot_delta_y_limit = counter_per_gait, counter_per_swing, control_dt, dx, dy
    p.default_foot_pos[:] = list(np.asarray(default_foot_pos, dtype=float).reshape(12))
    p.gait_counter_reset[:] = list(reset)
    return p


def update_plan(gp, movement_mode, gait_counter, gait_counter_speed, root_lin_vel, Rz, Rw, root_pos, root_lin_vel_d):
    """S/A1RobotControl.cpp:148-202 for one robot; returns (gait_counter, plan_contacts, target_rel, target_abs, target_world)."""
    gc = np.array(gait_counter, dtype=np.float64)
    pc = np.zeros(4, np.uint8); rel = np.zeros(12); ab = np.zeros(12); wo = np.zeros(12)
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    lib().orc_update_plan(C.byref(gp), C.c_int(int(movement_mode)), _p(gc), a(gait_counter_speed), a(root_lin_vel), a(Rz), a(Rw), a(root_pos),
                          a(root_lin_vel_d), _p(pc, C.c_uint8), _p(rel), _p(ab), _p(wo))
    return gc, pc, rel, ab, wo


def joint_torques(active, contacts, Jb, grf, f_kin, km, torques_gravity, joint_torques_prev):
    """S/A1RobotControl.cpp:289-319 for one robot (Jb = the four diagonal 3x3 blocks of j_foot, column-major)."""
    tau = np.array(joint_torques_prev, dtype=np.float64).reshape(12)
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    c = np.ascontiguousarray(contacts, dtype=np.uint8)
    lib().orc_joint_torques(C.c_int(int(active)), _p(c, C.c_uint8), a(Jb), a(grf), a(f_kin), a(km), a(torques_gravity), _p(tau))
    return tau


def contact_state():
    """zeroed N2b state of one robot (13 moving-window filters, early contacts, recent contact positions)"""
    lib().orc_contact_state_doubles.restype = C.c_int
    return np.zeros(lib().orc_contact_state_doubles())


def contact_terrain_step(state, gait_counter, plan_contacts, foot_force, foot_pos_abs, root_pos_z, pitch_d, counter_per_swing=120.0,
                         foot_force_low=30.0, use_terrain_adapt=1):
    """S/A1RobotControl.cpp:256-282, 566-582, 335-376 for one robot and one tick; state is updated in place.
    returns (contacts, foot_pos_recent_contact, terrain_angle, root_euler_d_pitch)"""
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    pc = np.ascontiguousarray(plan_contacts, dtype=np.uint8); ct = np.zeros(4, np.uint8); rec = np.zeros(12)
    ang = C.c_double(0.0); pd = C.c_double(float(pitch_d))
    lib().orc_contact_terrain_step(C.c_double(counter_per_swing), C.c_double(foot_force_low), C.c_int(int(use_terrain_adapt)), _p(state),
                                   a(gait_counter), _p(pc, C.c_uint8), a(foot_force), a(foot_pos_abs), C.c_double(float(root_pos_z)),
                                   _p(ct, C.c_uint8), _p(rec), C.byref(ang), C.byref(pd))
    return ct, rec, ang.value, pd.value


def swing_legs(Rz, foot_pos_abs, gait_counter, foot_pos_target_rel, foot_pos_start, rel_last, target_last, kp=(300.0, 400.0, 400.0),
               kd=(8.0, 8.0, 8.0), counter_per_swing=120.0, dt=0.0025):
    """S/A1RobotControl.cpp:204-254 for one robot; the three state arrays are updated in place.  returns (foot_pos_cur, foot_forces_kin)"""
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    cur = np.zeros(12); kin = np.zeros(12)
    lib().orc_swing_legs(C.c_double(counter_per_swing), C.c_double(dt), a(Rz), a(foot_pos_abs), a(gait_counter), a(foot_pos_target_rel), a(kp), a(kd),
                         _p(foot_pos_start), _p(rel_last), _p(target_last), _p(cur), _p(kin))
    return cur, kin


A1_RHO_FIX = np.array([[0.1805, 0.047, 0.0838, 0.21, 0.21], [0.1805, -0.047, -0.0838, 0.21, 0.21], [-0.1805, 0.047, 0.0838, 0.21, 0.21],
                       [-0.1805, -0.047, -0.0838, 0.21, 0.21]])  # S/GazeboA1ROS.cpp:76-93


def leg_state(joint_pos, joint_vel, Rw, root_pos, root_lin_vel, rho_fix=A1_RHO_FIX, rho_opt=np.zeros((4, 3))):
    """S/GazeboA1ROS.cpp:264-279 for one robot: dict of foot_pos_rel, Jb (4 blocks, column-major), foot_vel_rel, foot_pos_abs, foot_vel_abs,
    foot_pos_world, foot_vel_world"""
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    out = {k: np.zeros(36 if k == "Jb" else 12) for k in ("foot_pos_rel", "Jb", "foot_vel_rel", "foot_pos_abs", "foot_vel_abs", "foot_pos_world", "foot_vel_world")}
    lib().orc_leg_state(a(joint_pos), a(joint_vel), a(Rw), a(root_pos), a(root_lin_vel), a(rho_fix), a(rho_opt), *[_p(out[k]) for k in
                        ("foot_pos_rel", "Jb", "foot_vel_rel", "foot_pos_abs", "foot_vel_abs", "foot_pos_world", "foot_vel_world")])
    return out


def ekf_state():
    lib().orc_ekf_state_doubles.restype = C.c_int
    return np.zeros(lib().orc_ekf_state_doubles())


def ekf_step(state, dt, movement_mode, foot_force, Rw, imu_acc, imu_ang_vel, foot_pos_rel, foot_vel_rel, assume_flat_ground=1, device=False):
    """S/A1BasicEKF.cpp: init_state on the first call of a state, update_estimation afterwards.  returns (root_pos, root_lin_vel, estimated_contacts).
    device=False: the pinned restatement (multiply + add, the two solves as products with an explicit S^-1: what tests/test_ref_pin.py holds to the reference's compiled
    source); device=True: the device kernel's arithmetic (L D L' of S with [C Pbar | error_y] riding along) -- bit-comparable with a1mpc_ekf_update_batch, itself held to the
    pinned variant and to an 80-bit evaluation by the tests."""
    a = lambda v: _p(np.ascontiguousarray(v, dtype=np.float64))
    pos = np.zeros(3); vel = np.zeros(3); ec = np.zeros(4, np.uint8)
    (lib().orc_ekf_step_device if device else lib().orc_ekf_step)(_p(state), C.c_double(dt), C.c_int(int(assume_flat_ground)), C.c_int(int(movement_mode)), a(foot_force), a(Rw), a(imu_acc),
                       a(imu_ang_vel), a(foot_pos_rel), a(foot_vel_rel), _p(pos), _p(vel), _p(ec, C.c_uint8))
    return pos, vel, ec


def num_threads():
    return lib().orc_num_threads()
