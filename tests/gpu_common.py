"""Shared helpers of the GPU suite (engine factory, strided inputs of the general path, control-tick inputs): imported by the tests/test_gpu_*.py files."""
import numpy as np


def _engine(pkg, sc, max_batch, **osqp):
    cfg = pkg.make_config(sc["params"], sc["horizon"], **osqp)
    return pkg.Engine(cfg, max_batch=max_batch, device=0)


SETTINGS_CASES = [dict(scaling=0), dict(scaling=3), dict(alpha=1.0), dict(alpha=1.8), dict(rho=1.0), dict(rho=0.01, adaptive_rho=0),
                  dict(check_termination=10), dict(adaptive_rho_interval=50), dict(check_termination=10, adaptive_rho_interval=35),
                  dict(max_iter=30), dict(sigma=1e-4), dict(adaptive_rho_interval=0), dict(adaptive_rho_interval=0, check_termination=10),
                          dict(eps_abs=1e-5, eps_rel=1e-5), dict(adaptive_rho_tolerance=2.0)]


def tick_inputs(scen, rng, n):
    """random sensors / commands of one control tick (what test_device_pointer_tick_matches_host_pointer_tick feeds the chain)"""
    eul = rng.normal(0, 0.05, (n, 3)); eul[:, 2] = rng.uniform(-1, 1, n)
    return dict(joint_pos=np.tile([0.0, 0.8, -1.6], (n, 4)) + rng.normal(0, 0.1, (n, 12)), joint_vel=rng.normal(0, 1, (n, 12)),
                R_world=scen.rot_zyx(eul[:, 0], eul[:, 1], eul[:, 2]).reshape(n, 9), R_z=scen.rot_zyx(0 * eul[:, 0], 0 * eul[:, 0], eul[:, 2]).reshape(n, 9),
                root_euler=eul, root_ang_vel=rng.normal(0, 0.2, (n, 3)), imu_acc=np.array([0, 0, 9.81]) + rng.normal(0, 0.2, (n, 3)),
                imu_ang_vel=rng.normal(0, 0.2, (n, 3)), foot_force=rng.uniform(0, 120, (n, 4)), movement_mode=np.ones(n, np.uint8),
                mpc_active=(rng.random(n) < 0.9).astype(np.uint8), root_lin_vel_d=np.c_[rng.uniform(-0.4, 0.4, (n, 2)), np.zeros(n)],
                root_ang_vel_d=np.c_[np.zeros((n, 2)), rng.uniform(-0.4, 0.4, n)], root_pos_d_z=np.full(n, 0.3), gait_counter_speed=np.full((n, 4), 2.0),
                torques_gravity=rng.normal(0, 0.5, (n, 12)))


TICK_STATE = dict(gait_counter=4, foot_pos_start=12, foot_pos_rel_last_time=12, foot_pos_target_last_time=12, root_euler_d=3, joint_torques=12, root_pos=3,
                  root_lin_vel=3)
TICK_OUT_F64 = dict(foot_pos_rel=12, j_foot_blocks=36, foot_vel_rel=12, foot_pos_abs=12, foot_vel_abs=12, foot_pos_world=12, foot_vel_world=12, foot_pos_target_rel=12,
                    foot_pos_target_abs=12, foot_pos_target_world=12, foot_pos_cur=12, foot_forces_kin=12, foot_pos_recent_contact=12, terrain_angle=1, grf=12)


def _strided_inputs(scen, rng, h, nb, feet, cont):
    sc = scen.config3_random_flat(nb=nb, horizon=h)
    p = sc["params"]; foot = sc["foot"]; contact = sc["contact"]; fs = cs = 0
    if feet:
        vd = rng.uniform(-0.6, 0.6, (nb, 1, 1, 3))
        foot = (sc["foot"].reshape(nb, 1, 4, 3) - vd * p["dt"] * np.arange(h).reshape(1, h, 1, 1) * 40.0).reshape(nb, h * 12); fs = 12
    if cont:
        sw = rng.integers(0, h + 1, (nb, 4)); first = rng.integers(0, 2, (nb, 4))
        contact = np.where(np.arange(h).reshape(1, h, 1) < sw[:, None, :], first[:, None, :],
                1 - first[:, None, :]).astype(np.uint8).reshape(nb, h * 4); cs = 4
    return sc, np.ascontiguousarray(foot), fs, np.ascontiguousarray(contact), cs


def _oracle_update_ticks(oracle, pr, st, scs, carries):
    """one update-path tick of every robot b (its own carry) on the oracle"""
    n = len(scs["x0"])
    grf = np.zeros((n, 12)); it = np.zeros(n, np.int32); stt = np.zeros(n, np.int32)
    for b in range(n):
        o = oracle.mpc_solve_update(pr, st, scs["x0"][b], scs["xref"][b], scs["R"][b], scs["foot"][b], scs["contact"][b], carries[b])
        grf[b] = o["grf"]; it[b] = o["info"].iters; stt[b] = o["info"].status
    return grf, it, stt
