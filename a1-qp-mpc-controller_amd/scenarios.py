"""Synthetic input generators for the BASELINE.json configs.

Host-side only (numpy).  Every generator returns a dict of C-contiguous float64/uint8 arrays laid
out exactly as the C-ABI (include/a1mpc.h) takes them:
  x0 (nb,13)  xref (nb,13*h)  R (nb,9 row-major)  foot (nb,12 = 3x4 column-major)  contact (nb,4)
plus the parameter set to put into a1mpc_config.
"""
import numpy as np

NS, NU = 13, 12
GRAVITY_STATE = -9.8  # S/A1RobotControl.cpp:456

# parameter sets: config/gazebo_a1_mpc.yaml:6-72, hardware_a1_mpc.yaml:7-73, isaac_a1_mpc.yaml,
# S/test/test_mpc.cpp:18-60, defaults S/A1CtrlStates.h:40-60
_I = [0.0158533, 0, 0, 0, 0.0377999, 0, 0, 0, 0.0456542]
PARAM_SETS = {
    "gazebo": dict(mass=12.0, inertia=_I, q=[20, 10, 1, 0, 0, 420, .05, .05, .05, 30, 30, 10, 0], r=[1e-7] * 12,
                   foot=[[0.17, 0.15, -0.35], [0.17, -0.15, -0.35], [-0.17, 0.15, -0.35], [-0.17, -0.15, -0.35]]),
    "hardware": dict(mass=13.5, inertia=[0.0178533, 0, 0, 0, 0.0377999, 0, 0, 0, 0.0456542],
                     q=[150, 150, 50, 0, 0, 80, .2, .2, .2, .3, .3, .3, 0], r=[1e-2, 1e-2, 1e-3] * 4,
                     foot=[[0.17, 0.15, -0.3], [0.17, -0.15, -0.3], [-0.17, 0.15, -0.3], [-0.17, -0.15, -0.3]]),
    "isaac": dict(mass=12.0, inertia=_I, q=[100, 100, 50, 0, 0, 420, .01, .01, .05, 30, 30, 10, 0], r=[1e-7] * 12,
                  foot=[[0.24, 0.15, -0.35], [0.24, -0.15, -0.35], [-0.17, 0.15, -0.35], [-0.17, -0.15, -0.35]]),
    "test_mpc": dict(mass=15.0, inertia=_I, q=[1, 1, 1, 0, 0, 50, 0, 0, 1, 1, 1, 1, 0], r=[1e-6] * 12,
                     foot=[[0.17, 0.15, -0.35], [0.17, -0.15, -0.35], [-0.17, 0.15, -0.35], [-0.17, -0.15, -0.35]]),
    "ctrl_default": dict(mass=15.0, inertia=_I, q=[80, 80, 1, 0, 0, 270, 1, 1, 20, 20, 20, 20, 0],
                         r=[1e-5, 1e-5, 1e-6] * 4,
                         foot=[[0.17, 0.15, -0.35], [0.17, -0.15, -0.35], [-0.17, 0.15, -0.35], [-0.17, -0.15, -0.35]]),
}
MPC_CONSTANTS = dict(dt=0.0025, mu=0.3, fz_min=0.0, fz_max=180.0)  # S/A1RobotControl.cpp:462, S/ConvexMpc.cpp:8,223-224


def rot_zyx(roll, pitch, yaw):
    """R = Rz(yaw) Ry(pitch) Rx(roll)  (the convention S/utils/Utils.cpp:7-33 inverts). Vectorised -> (...,3,3)."""
    roll, pitch, yaw = np.broadcast_arrays(np.asarray(roll, float), np.asarray(pitch, float), np.asarray(yaw, float))
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.empty(roll.shape + (3, 3))
    R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp; R[..., 2, 1] = cp * sr; R[..., 2, 2] = cp * cr
    return R


def build_reference(h, dt, euler, pos, R, euler_d, lin_vel_d_body, ang_vel_d, pos_z_d):
    """x_ref exactly as S/A1RobotControl.cpp:470-488 builds it.  All inputs batched (nb, ...)."""
    nb = euler.shape[0]
    vw = np.einsum("bij,bj->bi", R, lin_vel_d_body)
    k = (np.arange(h) + 1.0)[None, :]
    xr = np.zeros((nb, h, NS))
    xr[:, :, 0] = euler_d[:, 0:1]
    xr[:, :, 1] = euler_d[:, 1:2]
    xr[:, :, 2] = euler[:, 2:3] + ang_vel_d[:, 2:3] * dt * k
    xr[:, :, 3] = pos[:, 0:1] + vw[:, 0:1] * dt * k
    xr[:, :, 4] = pos[:, 1:2] + vw[:, 1:2] * dt * k
    xr[:, :, 5] = np.asarray(pos_z_d).reshape(nb, 1)
    xr[:, :, 6:9] = ang_vel_d[:, None, :]
    xr[:, :, 9] = vw[:, 0:1]
    xr[:, :, 10] = vw[:, 1:2]
    xr[:, :, 11] = 0.0
    xr[:, :, 12] = GRAVITY_STATE
    return xr.reshape(nb, h * NS)


def pack_x0(euler, pos, ang_vel, lin_vel):
    """S/A1RobotControl.cpp:452-456."""
    nb = euler.shape[0]
    x0 = np.zeros((nb, NS))
    x0[:, 0:3] = euler; x0[:, 3:6] = pos; x0[:, 6:9] = ang_vel; x0[:, 9:12] = lin_vel; x0[:, 12] = GRAVITY_STATE
    return x0


def pack_tick(euler, pos, ang_vel, lin_vel, euler_d, lin_vel_d_body, ang_vel_d, pos_z_d):
    """The compact tick record of a1mpc_solve_batch_ticks (include/a1mpc.h): 22 doubles per QP."""
    nb = euler.shape[0]
    return np.ascontiguousarray(np.concatenate([euler, pos, ang_vel, lin_vel, euler_d, lin_vel_d_body, ang_vel_d,
                                                np.asarray(pos_z_d, float).reshape(nb, 1)], axis=1))


def _finish(params, horizon, x0, xref, R, foot, contact, **extra):
    out = dict(horizon=int(horizon), params=dict(params, **MPC_CONSTANTS),
               x0=np.ascontiguousarray(x0, dtype=np.float64), xref=np.ascontiguousarray(xref, dtype=np.float64),
               R=np.ascontiguousarray(R.reshape(-1, 9), dtype=np.float64),
               foot=np.ascontiguousarray(foot.reshape(-1, 12), dtype=np.float64),
               contact=np.ascontiguousarray(contact, dtype=np.uint8))
    out.update(extra)
    return out


def scenario_T(horizon=10):
    """Fixture T = the inputs of S/test/test_mpc.cpp:18-60 (stand, contacts FL+RL, cold start)."""
    p = PARAM_SETS["test_mpc"]
    euler = np.zeros((1, 3)); pos = np.array([[0.0, 0.0, 0.15]]); z = np.zeros((1, 3))
    R = rot_zyx(0.0, 0.0, 0.0)[None]
    x0 = pack_x0(euler, pos, z, z)
    # test_mpc.cpp:76-91 -- with all desired velocities zero x_ref == x0 at every step
    xref = np.tile(x0, (1, horizon))
    foot = np.array(p["foot"], dtype=float)[None]  # (1,4,3): leg-major == 3x4 column-major
    return _finish(p, horizon, x0, xref, R, foot, np.array([[1, 0, 1, 0]], dtype=np.uint8))


def scenario_stand(param_set="gazebo", horizon=10, height=0.3):
    """Analytic sanity case: 4 contacts, x_ref == x0  =>  f_z ~= m g / 4 per leg."""
    p = PARAM_SETS[param_set]
    euler = np.zeros((1, 3)); pos = np.array([[0.0, 0.0, height]]); z = np.zeros((1, 3))
    R = rot_zyx(0.0, 0.0, 0.0)[None]
    x0 = pack_x0(euler, pos, z, z)
    xref = build_reference(horizon, MPC_CONSTANTS["dt"], euler, pos, R, z, z, z, np.array([height]))
    foot = np.array(p["foot"], dtype=float)[None]
    return _finish(p, horizon, x0, xref, R, foot, np.ones((1, 4), dtype=np.uint8))


def _random_states(rng, nb, params, horizon, rpy_lim, z_rng, w_sig, v_sig, vd_lim, contact, pitch=None, v_zero=False,
                   foot_jitter=0.03, wz_d_lim=0.0):
    roll = rng.uniform(-rpy_lim[0], rpy_lim[0], nb)
    if pitch is None:
        pit = rng.uniform(-rpy_lim[1], rpy_lim[1], nb)
    else:
        pit = pitch[0] + rng.uniform(-pitch[1], pitch[1], nb)
    yaw = rng.uniform(-rpy_lim[2], rpy_lim[2], nb)
    euler = np.stack([roll, pit, yaw], 1)
    pos = np.stack([rng.normal(0, 1.0, nb), rng.normal(0, 1.0, nb), rng.uniform(z_rng[0], z_rng[1], nb)], 1)
    ang_vel = rng.normal(0, w_sig, (nb, 3))
    lin_vel = np.zeros((nb, 3)) if v_zero else rng.normal(0, v_sig, (nb, 3))
    R = rot_zyx(roll, pit, yaw)
    vd = np.stack([rng.uniform(-vd_lim[0], vd_lim[0], nb), rng.uniform(-vd_lim[1], vd_lim[1], nb), np.zeros(nb)], 1)
    wd = np.stack([np.zeros(nb), np.zeros(nb), rng.uniform(-wz_d_lim, wz_d_lim, nb) if wz_d_lim > 0 else np.zeros(nb)], 1)
    euler_d = np.zeros((nb, 3))
    if pitch is not None:
        euler_d[:, 1] = np.clip(pit, -0.5, 0.5)  # terrain adaptation writes +-terrain_angle, clamp 0.5 (Q6)
    x0 = pack_x0(euler, pos, ang_vel, lin_vel)
    xref = build_reference(horizon, MPC_CONSTANTS["dt"], euler, pos, R, euler_d, vd, wd, np.full(nb, 0.3))
    nominal = np.array(params["foot"], dtype=float)  # body frame (4,3)
    foot_body = nominal[None] + rng.uniform(-foot_jitter, foot_jitter, (nb, 4, 3))
    foot = np.einsum("bij,blj->bli", R, foot_body)  # foot_pos_abs = R * foot_pos_rel
    _random_states.last_tick = pack_tick(euler, pos, ang_vel, lin_vel, euler_d, vd, wd, np.full(nb, 0.3))
    return x0, xref, R, foot, contact


def config2_trot_sequence(nticks, seed=0xA1 + 2, horizon=10, param_set="gazebo"):
    """Config 2: trot, h=10, batch 1 -- `nticks` sequential ticks (contacts alternate 1001/0110 every 60 ticks,
    the reference's swing duration counter_per_swing/gait speed ~ S/A1CtrlStates.h:24-25)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    p = PARAM_SETS[param_set]
    nb = nticks
    euler = rng.normal(0, 0.02, (nb, 3)); pos = np.zeros((nb, 3)); pos[:, 2] = 0.3 + rng.normal(0, 0.01, nb)
    ang_vel = rng.normal(0, 0.1, (nb, 3)); lin_vel = rng.normal(0, 0.05, (nb, 3)); lin_vel[:, 0] += 0.3
    R = rot_zyx(euler[:, 0], euler[:, 1], euler[:, 2])
    vd = np.tile(np.array([[0.3, 0.0, 0.0]]), (nb, 1)); zero = np.zeros((nb, 3))
    x0 = pack_x0(euler, pos, ang_vel, lin_vel)
    xref = build_reference(horizon, MPC_CONSTANTS["dt"], euler, pos, R, zero, vd, zero, np.full(nb, 0.3))
    phase = (np.arange(nb) // 60) % 2
    contact = np.where(phase[:, None] == 0, np.array([[1, 0, 0, 1]]), np.array([[0, 1, 1, 0]])).astype(np.uint8)
    nominal = np.array(p["foot"], dtype=float)
    foot = np.einsum("bij,lj->bli", R, nominal)
    return _finish(p, horizon, x0, xref, R, foot, contact)


def config3_random_flat(nb=4096, seed=0xA1 + 3, horizon=10, param_set="gazebo"):
    """Config 3 (and 4 with horizon=16, nb=65536): randomized CoM states, flat terrain."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    p = PARAM_SETS[param_set]
    pat = np.array([[1, 1, 1, 1], [1, 0, 0, 1], [0, 1, 1, 0]], dtype=np.uint8)
    sel = rng.choice(3, size=nb, p=[0.5, 0.25, 0.25])
    x0, xref, R, foot, contact = _random_states(rng, nb, p, horizon, (0.15, 0.15, np.pi), (0.25, 0.32), 0.3, 0.3,
                                                (0.6, 0.3), pat[sel])
    return _finish(p, horizon, x0, xref, R, foot, contact, tick=_random_states.last_tick)


def config4_random_h16(nb=65536, seed=0xA1 + 4):
    return config3_random_flat(nb=nb, seed=seed, horizon=16)


def config5_divergent(nb=32768, seed=0xA1 + 5, horizon=20, param_set="gazebo"):
    """Config 5: all 15 non-empty contact patterns, pitch 0.5 +- 0.05 rad in x0/R/x_ref, v=0 with v_d up to 0.6."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    p = PARAM_SETS[param_set]
    code = rng.integers(1, 16, size=nb)
    contact = ((code[:, None] >> np.arange(4)[None, :]) & 1).astype(np.uint8)
    x0, xref, R, foot, contact = _random_states(rng, nb, p, horizon, (0.15, 0.0, np.pi), (0.25, 0.32), 0.3, 0.0,
                                                (0.6, 0.3), contact, pitch=(0.5, 0.05), v_zero=True)
    return _finish(p, horizon, x0, xref, R, foot, contact, tick=_random_states.last_tick)


def config1_balance_stand(param_set="gazebo"):
    """Config 1: one 12-var balance QP, stand, contacts 1111, root_acc = (0,0,m*9.8,0,0,0)."""
    p = PARAM_SETS[param_set]
    R = rot_zyx(0.0, 0.0, 0.0)
    foot = np.array(p["foot"], dtype=float)
    root_acc = np.array([[0, 0, p["mass"] * 9.8, 0, 0, 0]], dtype=float)
    return dict(params=p, root_acc=root_acc, R=R.reshape(1, 9).copy(), Rz=R.reshape(1, 9).copy(), foot=foot.reshape(1, 12).copy(),
                contact=np.ones((1, 4), dtype=np.uint8))


def balance_random(nb=256, seed=0xA1 + 1, param_set="gazebo"):
    """Randomised balance-QP inputs (h=1 analogue) for parity tests."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    p = PARAM_SETS[param_set]
    roll = rng.uniform(-0.15, 0.15, nb); pit = rng.uniform(-0.15, 0.15, nb); yaw = rng.uniform(-np.pi, np.pi, nb)
    R = rot_zyx(roll, pit, yaw); Rz = rot_zyx(0 * yaw, 0 * yaw, yaw)
    nominal = np.array(p["foot"], dtype=float)
    foot = np.einsum("bij,blj->bli", R, nominal[None] + rng.uniform(-0.03, 0.03, (nb, 4, 3)))
    root_acc = np.zeros((nb, 6))
    root_acc[:, 0:3] = rng.normal(0, 30, (nb, 3)); root_acc[:, 2] += p["mass"] * 9.8
    root_acc[:, 3:6] = rng.normal(0, 8, (nb, 3))
    code = rng.integers(1, 16, size=nb)
    contact = ((code[:, None] >> np.arange(4)[None, :]) & 1).astype(np.uint8)
    contact[: nb // 2] = 1
    return dict(params=p, root_acc=root_acc, R=np.ascontiguousarray(R.reshape(nb, 9)), Rz=np.ascontiguousarray(Rz.reshape(nb, 9)),
                foot=np.ascontiguousarray(foot.reshape(nb, 12)), contact=contact)
