"""Builds the product library liba1mpc.so (hand-written HIP for gfx950 + the C ABI of include/a1mpc.h).

hipcc cross-compiles without a GPU; the .so is written IN-TREE next to this file so it travels with the
source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "liba1mpc.so")
SOURCES = ["a1mpc_hip.hip", "a1mpc_solver.hpp", "a1mpc_tables.hpp", os.path.join("gfx950", "a1mpc_rowops.hpp")]
ARCH = "gfx950"


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build liba1mpc.so)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "a1mpc.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(CSRC, "gfx950"),
           "-I", CSRC, os.path.join(CSRC, "a1mpc_hip.hip"), "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
