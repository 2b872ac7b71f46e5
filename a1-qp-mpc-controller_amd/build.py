"""Builds the product library liba1mpc.so (hand-written HIP for gfx950 + the C ABI of include/a1mpc.h).

hipcc cross-compiles without a GPU; the .so is written IN-TREE next to this file so it travels with the
source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import shutil
import subprocess
import tempfile

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
    """hipcc -> liba1mpc.so, then the DPP hazard check of the generated gfx950 assembly (isa_check.py): a library whose inline-asm
    DPP chains read a register too early after a VALU write is deleted again and the build fails."""
    if not force and not needs_build():
        return LIB_PATH
    from . import isa_check
    with tempfile.TemporaryDirectory(prefix="a1mpc_build_") as tmp:  # -save-temps writes the listings into the working directory
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-save-temps", "-I", os.path.join(CSRC, "gfx950"),
               "-I", CSRC, os.path.join(CSRC, "a1mpc_hip.hip"), "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=tmp)
        listings = glob.glob(os.path.join(tmp, f"*{ARCH}*.s"))
        if not listings:
            raise RuntimeError("hipcc -save-temps left no gfx950 listing to check")
        bad = [h for l in listings for h in isa_check.dpp_hazards(l)]
        if os.environ.get("A1MPC_SLIM") is None:
            bad += isa_check.coverage_gaps(listings)   # the check must have seen the kernels it exists for (no fail-open)
    if bad:
        os.remove(LIB_PATH)
        raise RuntimeError("DPP read hazards in the generated code (library removed):\n" + "\n".join(bad[:20]))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
