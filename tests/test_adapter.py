"""The C++ adapter with the reference's ConvexMpc / compute_grf interface (include/a1mpc_convex_mpc.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_adapter.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_adapter")


def _build(pkg):
    pkg.build.build()
    libdir = os.path.dirname(pkg.build.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir, "-la1mpc",
                           f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])


def test_adapter_compiles_and_links_against_the_c_abi(pkg):
    _build(pkg)  # template instantiation with a non-Eigen matrix shim + link against liba1mpc.so
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_adapter_runs_fixture_T(pkg):
    if not os.path.exists(EXE):
        _build(pkg)
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    print(out.stdout)
    assert out.returncode == 0 and "ADAPTER_OK" in out.stdout, out.stdout + out.stderr
