#!/usr/bin/env python3
"""Runs ON THE GPU BOX: parity soak of the EKF kernel (N4c) on fresh seeds -- EVERY robot of every tick against the oracle's device variant bit for bit and against the
pinned restatement within 1e-9 (reported: the explicit inverse of the pinned variant is the less accurate side, tests/test_oracle.py); both ground assumptions, standing fleets (movement_mode 0), force extremes, a reset in the middle, odd batch sizes.
usage: python tools/ekf_soak.py [seed]   (test infrastructure: the oracle is the checker here)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("a1-qp-mpc-controller_amd")
from oracle import oracle
scen = pkg.scenarios
seed = int(sys.argv[1]) if len(sys.argv) > 1 else int(time.time()) % 100000
rng = np.random.default_rng(seed)
base = np.array([0.18, 0.13, -0.3, 0.18, -0.13, -0.3, -0.18, 0.13, -0.3, -0.18, -0.13, -0.3])
cfg = pkg.make_config(scen.PARAM_SETS["gazebo"] | scen.MPC_CONSTANTS, 10)
total = 0; worst_pin = 0.0
for n, ticks, flat, kind in ((1537, 40, 1, "walk"), (2049, 30, 0, "walk"), (515, 60, 1, "stand"), (1023, 30, 1, "extremes"), (33, 120, 0, "walk"), (1, 80, 1, "walk")):
    dev = [oracle.ekf_state() for _ in range(n)]; pin = [oracle.ekf_state() for _ in range(n)]
    with pkg.Engine(cfg, n, 0) as eng:
        for t in range(ticks):
            if kind == "stand": mm = np.zeros(n, np.uint8)
            else: mm = np.where(rng.random(n) < 0.8, 1, 0).astype(np.uint8) if t > 2 else np.zeros(n, np.uint8)
            yaw = rng.uniform(-3.1, 3.1, n); eul = rng.normal(0, 0.08, (n, 2)); R = scen.rot_zyx(eul[:, 0], eul[:, 1], yaw).reshape(n, 9)
            fk = base + rng.normal(0, 0.02, (n, 12)); fv = rng.normal(0, 0.5, (n, 12)); acc = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.5, (n, 3))
            w = rng.normal(0, 0.5, (n, 3)); ff = rng.uniform(0, 160, (n, 4))
            if kind == "extremes": ff = rng.choice([0.0, 1e-300, 49.999999, 50.0, 100.0, 1e3, 1e6], (n, 4))
            if t == ticks // 2 and kind == "walk" and n > 1:   # a reset in the middle: every filter starts over
                eng.reset_ekf_state(); dev = [oracle.ekf_state() for _ in range(n)]; pin = [oracle.ekf_state() for _ in range(n)]
            pos, vel, ec = eng.ekf_update(0.0025, mm, ff, R, acc, w, fk, fv, assume_flat_ground=flat)
            for b in range(n):
                p_o, v_o, e_o = oracle.ekf_step(dev[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], assume_flat_ground=flat, device=True)
                assert np.array_equal(pos[b], p_o) and np.array_equal(vel[b], v_o) and (ec[b] == e_o).all(), (seed, n, kind, t, b, pos[b] - p_o, vel[b] - v_o)
                p_p, v_p, e_p = oracle.ekf_step(pin[b], 0.0025, mm[b], ff[b], R[b], acc[b], w[b], fk[b], fv[b], assume_flat_ground=flat)
                d = max(np.abs(pos[b] - p_p).max(), np.abs(vel[b] - v_p).max()); worst_pin = max(worst_pin, d)
                assert d <= 1e-9 and (ec[b] == e_p).all(), (seed, n, kind, t, b, d)   # (1e-9: what holds the pinned restatement to the reference; the GPU tests hold 1e-10 on their seeds)
            total += n
    print(f"n = {n:5d}  {ticks:3d} ticks  flat = {flat}  {kind:8s}: every robot of every tick bit for bit the oracle's device variant", flush=True)
print(f"seed {seed}: {total} robot-ticks, 0 mismatches against the device variant; worst distance to the pinned restatement {worst_pin:.3g} (bound 1e-9; 1e-10 in the GPU tests)")
