"""The caller-side rows (N2a update_plan, N2b contacts / terrain, N4a swing legs, N4b leg kinematics, N4c EKF) under RANDOM configurations -- gait constants, contact thresholds,
PD gains, swing length, control dt, flat-ground flag, kinematic parameters -- robot by robot against the oracle (GPU).  The gated suite runs them at the reference's constants.
usage: soak_caller_side.py [first_seed [count [robots]]]"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle(); E = pkg.engine; scen = pkg.scenarios
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 800; cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 20; n = int(sys.argv[3]) if len(sys.argv) > 3 else 96
cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
bad = {k: 0 for k in ("plan", "contact", "swing", "leg", "ekf")}; worst = {k: 0.0 for k in bad}; checks = 0
for seed in range(lo, lo + cnt):
    rng = np.random.default_rng(seed)
    # ---- random constants
    cps = float(rng.choice([60.0, 100.0, 120.0, 150.0])); cpg = 2 * cps; cdt = float(rng.choice([0.0025, 0.002, 0.004]))
    dfp = np.array([0.17, 0.15, -0.35, 0.17, -0.15, -0.35, -0.17, 0.15, -0.35, -0.17, -0.15, -0.35]) * rng.uniform(0.8, 1.2, 12)
    dx, dy = float(rng.uniform(0.05, 0.2)), float(rng.uniform(0.05, 0.2)); reset = (0.0, cps, cps, 0.0)
    gait = E.GaitConfig(); E.load_library().a1mpc_default_gait_config(C.byref(gait))
    gait.counter_per_gait, gait.counter_per_swing, gait.control_dt, gait.foot_delta_x_limit, gait.foot_delta_y_limit = cpg, cps, cdt, dx, dy
    gait.default_foot_pos[:] = list(dfp); gait.gait_counter_reset[:] = list(reset)
    gp = orc.gait_params(dfp, cpg, cps, cdt, dx, dy, reset)
    cc = E.ContactConfig(); cc.counter_per_swing = cps; cc.foot_force_low = float(rng.uniform(10, 60)); cc.use_terrain_adapt = int(rng.integers(0, 2))
    kp = tuple(rng.uniform(100, 500, 3)); kd = tuple(rng.uniform(2, 15, 3)); ekf_dt = cdt; flat = int(rng.integers(0, 2))
    rho_fix = E.Engine.A1_RHO_FIX * rng.uniform(0.9, 1.1, (4, 5)) if hasattr(E.Engine, "A1_RHO_FIX") else None; rho_opt = rng.normal(0, 0.01, (4, 3))
    ct_states = [orc.contact_state() for _ in range(n)]; ekf_states = [orc.ekf_state() for _ in range(n)]
    sw_g = [np.zeros((n, 12)) for _ in range(3)]; sw_o = [np.zeros((n, 12)) for _ in range(3)]
    gcs = rng.uniform(0, cpg, (n, 4)); pitch_g = np.zeros(n); pitch_o = np.zeros(n)
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(40):
            yaw = rng.uniform(-3, 3, n); eul = rng.normal(0, 0.1, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9); Rz = scen.rot_zyx(0 * yaw, 0 * yaw, yaw).reshape(n, 9)
            mm = (rng.random(n) < 0.85).astype(np.uint8); spd = rng.choice([1.0, 1.5, 2.0], size=(n, 4)); v = rng.normal(0, 0.5, (n, 3)); vd = rng.normal(0, 0.6, (n, 3)); pos = rng.normal(0, 1.0, (n, 3)); pos[:, 2] = rng.choice([0.3, 0.05], n, p=[0.9, 0.1])
            up = eng.update_plan(mm, gcs, spd, v, Rz, R, pos, vd, gait=gait)
            q = rng.uniform(-1.0, 1.0, (n, 12)); qd = rng.normal(0, 2, (n, 12))
            leg = eng.leg_state(q, qd, R, pos, v, rho_fix=rho_fix, rho_opt=rho_opt)
            ff = rng.uniform(0, 100, (n, 4))
            ctr = eng.contact_terrain(up["gait_counter"], up["plan_contacts"], ff, leg["foot_pos_abs"], pos[:, 2], pitch_g, cfg=cc); pitch_g = ctr["root_euler_d_pitch"]
            cur, kin = eng.swing_legs(Rz, leg["foot_pos_abs"], up["gait_counter"], up["foot_pos_target_rel"], *sw_g, kp=kp, kd=kd, counter_per_swing=cps, dt=cdt)
            acc = np.array([0, 0, 9.81]) + rng.normal(0, 0.3, (n, 3)); w = rng.normal(0, 0.3, (n, 3))
            ep, ev, ec = eng.ekf_update(ekf_dt, mm, ff, R, acc, w, leg["foot_pos_rel"], leg["foot_vel_rel"], assume_flat_ground=flat)
            for b in range(n):
                g2, pc, rel, ab, wo = orc.update_plan(gp, mm[b], gcs[b], spd[b], v[b], Rz[b], R[b], pos[b], vd[b])
                ok = (up["gait_counter"][b] == g2).all() and (up["plan_contacts"][b] == pc).all() and (up["foot_pos_target_rel"][b] == rel).all() and (up["foot_pos_target_abs"][b] == ab).all() and (up["foot_pos_target_world"][b] == wo).all()
                bad["plan"] += int(not ok)
                lo_ = orc.leg_state(q[b], qd[b], R[b], pos[b], v[b], **({"rho_fix": rho_fix} if rho_fix is not None else {}), rho_opt=rho_opt)
                dl = max(np.abs(leg[k][b] - lo_[k]).max() for k in lo_); worst["leg"] = max(worst["leg"], dl); bad["leg"] += int(dl > 1e-12)
                ct, rec, ang, pitch_o[b] = orc.contact_terrain_step(ct_states[b], g2, pc, ff[b], leg["foot_pos_abs"][b], pos[b, 2], pitch_o[b], cps, cc.foot_force_low, cc.use_terrain_adapt)
                okc = (ctr["contacts"][b] == ct).all() and (ctr["foot_pos_recent_contact"][b] == rec).all() and abs(ctr["terrain_angle"][b] - ang) <= 1e-13 and abs(pitch_g[b] - pitch_o[b]) <= 1e-13
                bad["contact"] += int(not okc)
                c_o, k_o = orc.swing_legs(Rz[b], leg["foot_pos_abs"][b], g2, rel, sw_o[0][b], sw_o[1][b], sw_o[2][b], kp=kp, kd=kd, counter_per_swing=cps, dt=cdt)
                oks = (cur[b] == c_o).all() and (sw_g[0][b] == sw_o[0][b]).all() and (sw_g[1][b] == sw_o[1][b]).all() and np.abs(sw_g[2][b] - sw_o[2][b]).max() <= 1e-15
                dk = float(np.abs(kin[b] - k_o).max()); worst["swing"] = max(worst["swing"], dk); bad["swing"] += int(not oks or dk > 1e-8); sw_o[2][b] = sw_g[2][b]
                p_o, v_o, e_o = orc.ekf_step(ekf_states[b], ekf_dt, mm[b], ff[b], R[b], acc[b], w[b], leg["foot_pos_rel"][b], leg["foot_vel_rel"][b], assume_flat_ground=flat)
                de = max(np.abs(ep[b] - p_o).max(), np.abs(ev[b] - v_o).max()); worst["ekf"] = max(worst["ekf"], float(de)); bad["ekf"] += int(de > 1e-9 or not (ec[b] == e_o).all())
                checks += 1
            gcs = up["gait_counter"]
    print(seed, "cps %.0f dt %.4f flat %d terrain_adapt %d" % (cps, cdt, flat, cc.use_terrain_adapt), "mismatching so far", bad, "worst", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
print("TOTAL", checks, "robot-ticks x 5 rows under", cnt, "random configurations: mismatching", bad, "worst deviations", {k: "%.2e" % v for k, v in worst.items()})
