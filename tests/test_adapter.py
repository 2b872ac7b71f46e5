"""The C++ side of the boundary: include/a1mpc_dropin.hpp (the reference's ConvexMpc / compute_grf interface over the C ABI) and the C++
latency harness.  The programs are built by tests/cpp/Makefile -- the two that include the reference's headers only where /root/reference
exists (this container); the binaries travel to the GPU box with the snapshot."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build(pkg):
    pkg.build.build()
    import ref as REF
    REF.build()
    subprocess.check_call(["make", "-C", CPP, "all"], stdout=subprocess.DEVNULL)


def test_dropin_programs_compile_and_link(pkg):
    """a1mpc_dropin.hpp instantiated on the reference's A1CtrlStates next to the reference's own A1RobotControl (test_dropin), the reference's
    S/test/test_mpc.cpp compiled UNMODIFIED against the drop-in ConvexMpc (test_mpc_dropin), and the C-ABI latency harness."""
    _build(pkg)
    have_ref = os.path.isdir("/root/reference/src/a1_cpp/src")
    for exe in ("latency_harness",) + (("test_dropin", "test_mpc_dropin") if have_ref else ()):
        assert os.path.exists(os.path.join(CPP, exe)), exe


def _run(exe, *args, timeout=300):
    path = os.path.join(CPP, exe)
    if not os.path.exists(path):
        pytest.skip(f"{exe} was not built (needs /root/reference at build time)")
    return subprocess.run([path, *args], capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_dropin_equals_reference_classes():
    """ConvexMpcGpu members (hessian / gradient / lb / ub formed on the GPU, per-step feet, a separate A_c yaw) == the reference's ConvexMpc;
    ComputeGrfGpu::compute_grf == A1RobotControl::compute_grf over warm-started MPC ticks with the terrain block, and on the balance branch."""
    out = _run("test_dropin")
    print(out.stderr)
    assert out.returncode == 0 and "DROPIN_OK" in out.stderr, out.stderr


@pytest.mark.gpu
def test_reference_test_mpc_runs_on_the_dropin():
    """S/test/test_mpc.cpp as written, with ConvexMpc := a1mpc::ConvexMpcGpu: prints the forces of fixture T (FL = RL ~ (0, -12.78, 42.61) N at
    OSQP's default tolerances, FR = RR ~ 0)."""
    out = _run("test_mpc_dropin")
    print(out.stdout)
    assert out.returncode == 0, out.stderr
    rows = [[float(v) for v in ln.split()] for ln in out.stdout.strip().splitlines()[:3]]
    assert abs(rows[2][0] - 42.6055) < 1e-3 and abs(rows[2][2] - 42.6055) < 1e-3 and abs(rows[1][0] + 12.782) < 1e-3
    assert abs(rows[2][1]) < 1e-3 and abs(rows[2][3]) < 1e-3


@pytest.mark.gpu
def test_cpp_latency_harness_meets_the_tick():
    out = _run("latency_harness", "2000")
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    print(r)
    assert r["not_solved"] == 0 and r["p99_ms"] < 2.5
