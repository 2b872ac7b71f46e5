#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle (oracle/a1mpc_oracle.c), in this container.

The reference pins no expected values (S/test/test_mpc.cpp:157-161 only prints) and cannot be built here (Eigen,
OsqpEigen and ROS are absent), so these vectors are the oracle's own outputs, PINNED so that (a) an accidental change
of the oracle, the scenario generators or the compiler flags is caught, (b) the GPU box can check the HIP path
against numbers produced elsewhere.  Every `exact` solution in here is certified independently by
tests/test_oracle.py through the KKT conditions of a numpy re-formation of the QP (tests/ref_numpy.py).

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

O = g.load_oracle(); O.build()
S = g.load_package().scenarios


def solve(sc, st):
    p = sc["params"]
    pr = O.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    return O.mpc_solve_batch(pr, st, sc["x0"], sc["xref"], sc["R"], sc["foot"], sc["contact"], want_u=True)


CASES = {
    "T_test_mpc": lambda: S.scenario_T(),
    "stand_gazebo": lambda: S.scenario_stand("gazebo"),
    "stand_isaac": lambda: S.scenario_stand("isaac"),
    "stand_hardware": lambda: S.scenario_stand("hardware"),
    "config3_h10": lambda: S.config3_random_flat(nb=16),
    "config4_h16": lambda: S.config4_random_h16(nb=8),
    "config5_h20": lambda: S.config5_divergent(nb=8),
    "config3_hardware_weights": lambda: S.config3_random_flat(nb=8, param_set="hardware"),
}

if __name__ == "__main__":
    for name, gen in CASES.items():
        sc = gen()
        d = solve(sc, O.default_settings()); e = solve(sc, O.exact_settings())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), horizon=sc["horizon"], x0=sc["x0"], xref=sc["xref"], R=sc["R"],
                            foot=sc["foot"], contact=sc["contact"],
                            default_u=d["u"], default_grf=d["grf"], default_iters=d["iters"], default_status=d["status"],
                            exact_u=e["u"], exact_grf=e["grf"], exact_iters=e["iters"], exact_status=e["status"])
        print(name, "default iters", d["iters"].tolist(), "exact iters", e["iters"].tolist())
    # balance QP (config 1 + randomised)
    sc = S.balance_random(16)
    qp, st = O.default_qp_params(), O.default_settings()
    f = []; it = []
    for b in range(16):
        r = O.balance_solve(qp, st, sc["root_acc"][b], sc["R"][b], sc["Rz"][b], sc["foot"][b], sc["contact"][b])
        f.append(r["f_world"]); it.append(r["info"].iters)
    np.savez_compressed(os.path.join(HERE, "balance_random.npz"), **{k: sc[k] for k in ("root_acc", "R", "Rz", "foot", "contact")},
                        default_f=np.array(f), default_iters=np.array(it))
    # warm-started tick sequence (config 2)
    sc = S.config2_trot_sequence(12)
    p = sc["params"]
    pr = O.mpc_params(sc["horizon"], p["dt"], p["mu"], p["fz_min"], p["fz_max"], p["q"], p["r"], p["mass"], p["inertia"])
    st = O.default_settings(warm_start=1)
    wx = np.zeros(120); wy = np.zeros(200); rho = None; us = []; its = []
    for t in range(12):
        r = O.mpc_solve(pr, st, sc["x0"][t], sc["xref"][t], sc["R"][t], sc["foot"][t], sc["contact"][t], warm_x=wx, warm_y=wy, warm_rho=rho)
        wx, wy, rho = r["warm_x"], r["warm_y"], r["rho"]; us.append(r["u"]); its.append(r["info"].iters)
    np.savez_compressed(os.path.join(HERE, "config2_warm_sequence.npz"), horizon=10, x0=sc["x0"], xref=sc["xref"], R=sc["R"], foot=sc["foot"],
                        contact=sc["contact"], u=np.array(us), iters=np.array(its))
    print("warm sequence iters", its)
