#!/usr/bin/env python3
"""HBM roofline of the element-wise caller-side kernels (N2a update_plan, N3 joint torques, N2b contact/terrain, N4a / N4b / N4c): algorithmic bytes per robot / kernel time
(HIP events inside the library, a1mpc_last_kernel_ms).  usage: elementwise_probe.py [robots, default 65536].  A working set inside the 256 MiB Infinity Cache measures
cache bandwidth, not HBM (MI355X_MICROARCH.md: scale past L3 before reading a bandwidth): run with >= 524288 robots for the HBM figures (A1_SKIP_N2B=1 skips the
70-tick N2b sequence, which tools/ubench/n2b_bench measures at that size).  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); scen = pkg.scenarios
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
out = {}
with pkg.Engine(cfg, n, 0) as eng:
    yaw = rng.uniform(-3, 3, n); R = scen.rot_zyx(0 * yaw, 0 * yaw, yaw).reshape(n, 9)
    args = ((rng.random(n) < 0.8).astype(np.uint8), rng.uniform(0, 240, (n, 4)), np.full((n, 4), 2.0), rng.normal(0, 0.5, (n, 3)), R, R, rng.normal(0, 1, (n, 3)), rng.normal(0, 0.5, (n, 3)))
    ms = []
    for _ in range(5):
        eng.update_plan(*args); ms.append(eng.last_kernel_ms())
    b = 8 * (4 + 4 + 3 + 9 + 9 + 3 + 3) + 1 + 8 * (4 + 36) + 4   # in + out per robot
    out["N2a update_plan"] = {"kernel_ms": float(np.median(ms[1:])), "bytes_per_robot": b, "GB_per_s": n * b / (float(np.median(ms[1:])) * 1e-3) / 1e9}
    Jb = rng.normal(0, 0.2, (n, 36)); Jb[:, [0, 4, 8, 9, 13, 17, 18, 22, 26, 27, 31, 35]] += 0.3
    a3 = (np.ones(n, np.uint8), (rng.random((n, 4)) < 0.5).astype(np.uint8), Jb, rng.normal(0, 40, (n, 12)), rng.normal(0, 20, (n, 12)), np.array([0.1, 0.1, 0.04]),
          rng.normal(0, 1, (n, 12)), np.zeros((n, 12)))
    ms = []
    for _ in range(5):
        eng.joint_torques(*a3); ms.append(eng.last_kernel_ms())
    b = 8 * (36 + 12 + 12 + 12 + 12) + 5 + 8 * 12
    out["N3 joint_torques"] = {"kernel_ms": float(np.median(ms[1:])), "bytes_per_robot": b, "GB_per_s": n * b / (float(np.median(ms[1:])) * 1e-3) / 1e9}
    def n2b(e, nn):
        # robots' gait counters start at random phases, so their filters' ring cursors disagree from the first tick on (the general case: early contacts part them anyway)
        ms = []; legs = []
        gcs = rng.uniform(0, 240, (nn, 4)); pitch = np.zeros(nn)
        for t in range(70):
            gcs = np.fmod(gcs + 2.0, 240.0)
            o = e.contact_terrain(gcs, (gcs <= 120).astype(np.uint8), rng.uniform(0, 80, (nn, 4)), rng.normal(0, 0.2, (nn, 12)), np.full(nn, 0.3), pitch); pitch = o["root_euler_d_pitch"]
            ms.append(e.last_kernel_ms()); legs.append(float(o["contacts"].sum()) / nn)
        # per tick and robot: inputs 4+4+12+1+1 doubles + 4 bytes, outputs 12+1+1 doubles + 4 bytes, state: the 384-byte record read, per leg in contact its 64-byte header
        # written and one ring sector read (32) and written (24), the robot block written (recent 96 + terrain header 24), one word of the terrain ring read and written
        la = float(np.mean(legs[-5:]))
        b = 8 * 22 + 4 + 8 * 14 + 4 + 384 + la * (64 + 32 + 24) + 120 + 16
        t = float(np.median(ms[-5:]))
        return {"kernel_ms": t, "robots": nn, "legs_in_contact_per_robot": la, "bytes_per_robot": b, "GB_per_s": nn * b / (t * 1e-3) / 1e9}
    if os.environ.get("A1_SKIP_N2B") != "1":
        out["N2b contact_terrain (one fused kernel since round 3; random filter phases, windows full)"] = n2b(eng, n)
    # N4b leg kinematics, N4a swing legs, N4c EKF
    q = rng.uniform(-1, 1, (n, 12)); qd = rng.normal(0, 2, (n, 12)); pos = rng.normal(0, 1, (n, 3)); vel = rng.normal(0, 1, (n, 3))
    ms = []
    for _ in range(5):
        leg = eng.leg_state(q, qd, R, pos, vel); ms.append(eng.last_kernel_ms())
    b = 8 * (12 + 12 + 9 + 3 + 3) + 8 * (36 + 6 * 12)
    out["N4b leg_state"] = {"kernel_ms": float(np.median(ms[1:])), "bytes_per_robot": b, "GB_per_s": n * b / (float(np.median(ms[1:])) * 1e-3) / 1e9}
    stt = [np.zeros((n, 12)) for _ in range(3)]; ms = []
    for _ in range(5):
        eng.swing_legs(R, leg["foot_pos_abs"], rng.uniform(0, 240, (n, 4)), leg["foot_pos_rel"], *stt); ms.append(eng.last_kernel_ms())
    b = 8 * (9 + 12 + 4 + 12 + 36) + 8 * (36 + 24)
    out["N4a swing_legs"] = {"kernel_ms": float(np.median(ms[1:])), "bytes_per_robot": b, "GB_per_s": n * b / (float(np.median(ms[1:])) * 1e-3) / 1e9}
    ms = []
    for t in range(6):
        eng.ekf_update(0.0025, np.ones(n, np.uint8), rng.uniform(0, 150, (n, 4)), R, rng.normal(0, 1, (n, 3)) + [0, 0, 9.81], rng.normal(0, 0.2, (n, 3)), leg["foot_pos_rel"], leg["foot_vel_rel"])
        ms.append(eng.last_kernel_ms())
    b = 8 * (4 + 9 + 3 + 3 + 12 + 12) + 1 + 2 * 8 * 343 + 8 * 6 + 4   # inputs + state read and written + outputs
    fl = 2 * (18 * 18 * 2 + 28 * 18 * 2 + 28 * 28 * 28 + 28 * 28 + 28 * 4 + 18 * 28 + 18 * 28 * 18 + 18 * 18 * 18)  # products + the in-place 28 x 28 inverse (round 3; the 47-wide tableau until then: 28 * 28 * 47) + S^-1 e, S^-1 C: flops per robot
    out["N4c ekf (init + update kernels)"] = {"kernel_ms": float(np.median(ms[2:])), "bytes_per_robot": b, "GB_per_s": n * b / (float(np.median(ms[2:])) * 1e-3) / 1e9,
                                              "flops_per_robot": fl, "TFLOP_per_s": n * fl / (float(np.median(ms[2:])) * 1e-3) / 1e12}
if os.environ.get("A1_N2B_LARGE", "1") != "0" and os.environ.get("A1_SKIP_N2B") != "1":
    with pkg.Engine(cfg, 8 * n, 0) as big:
        out["N2b contact_terrain, 8 x the robots"] = n2b(big, 8 * n)
for v in out.values():
    v["frac_of_achievable_6300_GB_per_s"] = v["GB_per_s"] / 6300.0
    v["working_set_MB"] = v["bytes_per_robot"] * (v.get("robots", n)) / 1e6
print(json.dumps({"robots": n, "peak_GB_per_s": 8000, "achievable_GB_per_s": 6300, "infinity_cache_MB": 256, "kernels": out}))
