"""N2b on the CPU: the product's contact / terrain kernel text compiled for the host (tests/emu/n2b_host.py) stepped against the oracle -- the shipped
arithmetic AND the shipped state layout (records + sector-sized ring slots), window wrap of both filter lengths included.  The same sequence runs on the
GPU in tests/test_gpu_caller_side.py::test_contact_terrain_N2b_sequence."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import n2b_host


def test_n2b_kernel_text_on_the_host_matches_the_oracle(oracle):
    rng = np.random.default_rng(21)
    n, ticks = 24, 260   # terrain window 100 and leg window 60 both wrap; max_batch > n: the ring regions start behind max_batch records
    H = n2b_host.HostN2b(40)
    states = [oracle.contact_state() for _ in range(n)]
    pg = np.zeros(n); po = np.zeros(n)
    base = np.outer([0.2, 0.2, -0.2, -0.2], [1.0, 0.0, 0.3]).reshape(12) + np.outer([1, -1, 1, -1], [0.0, 0.13, 0.0]).reshape(12)
    gcs = rng.uniform(0, 240, (n, 4))
    for t in range(ticks):
        gcs = np.fmod(gcs + 2.0, 240.0); plan = (gcs <= 120).astype(np.uint8); ff = rng.uniform(0, 80, (n, 4))
        foot = base + rng.normal(0, 0.03, (n, 12)) + np.tile([0.0, 0.0, -0.3], 4); z = np.where(rng.random(n) < 0.9, 0.3, 0.05)
        out = H.tick(gcs, plan, ff, foot, z, pg); pg = out["root_euler_d_pitch"]
        for b in range(n):
            ct, rec, ang, po[b] = oracle.contact_terrain_step(states[b], gcs[b], plan[b], ff[b], foot[b], z[b], po[b])
            assert (out["contacts"][b] == ct).all() and (out["foot_pos_recent_contact"][b] == rec).all(), (t, b)
            assert out["terrain_angle"][b] == ang and pg[b] == po[b], (t, b)   # same libm on the host: exact (1e-13 on the GPU, whose acos is the device library's)


def test_n2b_terrain_only_entry_shares_the_terrain_filter(oracle):
    """recent_in given (a1mpc_terrain_batch): the leg filters are not touched, the terrain filter advances exactly as in the full entry fed the same positions"""
    rng = np.random.default_rng(5)
    n = 8
    A = n2b_host.HostN2b(n); B = n2b_host.HostN2b(n)
    pa = np.zeros(n); pb = np.zeros(n)
    gcs = rng.uniform(0, 240, (n, 4))
    for t in range(130):
        gcs = np.fmod(gcs + 2.0, 240.0); plan = (gcs <= 120).astype(np.uint8); ff = rng.uniform(0, 80, (n, 4)); foot = rng.normal(0, 0.2, (n, 12)); z = np.full(n, 0.3)
        oa = A.tick(gcs, plan, ff, foot, z, pa); pa = oa["root_euler_d_pitch"]
        ob = B.tick(gcs, plan, ff, foot, z, pb, recent_in=oa["foot_pos_recent_contact"]); pb = ob["root_euler_d_pitch"]
        assert np.array_equal(oa["terrain_angle"], ob["terrain_angle"]) and np.array_equal(pa, pb)
