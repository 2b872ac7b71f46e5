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


def source_hash():
    """sha256 (16 hex digits) over the library's sources: compiled into the library (a1mpc_build_info) so that a shipped liba1mpc.so can be matched to the sources
    beside it -- file times do not survive a snapshot"""
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "a1mpc.h")]:
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def library_hash():
    """the source hash the existing library was compiled from, or None (missing library / a build from before the hash existed).
    Read from the file's bytes (the string a1mpc_build_info() returns), NOT by loading the library: glibc matches loaded libraries by name, so a
    CDLL here would pin the old mapping and a later CDLL of the rebuilt file at the same path would silently return the pre-build code."""
    import re
    if not os.path.exists(LIB_PATH):
        return None
    try:
        with open(LIB_PATH, "rb") as f:
            m = re.search(rb"sources ([0-9a-f]{16}) arch " + ARCH.encode(), f.read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def needs_build():
    return library_hash() != source_hash()


last_build = {}   # what the last build() call did: {"compiled": bool, "source_hash": ..., "seconds": ...} (reported by __graft_entry__.build)


def build(force=False, verbose=False):
    """hipcc -> liba1mpc.so, then the DPP hazard check of the generated gfx950 assembly (isa_check.py): a library whose inline-asm
    DPP chains read a register too early after a VALU write is deleted again and the build fails."""
    import time
    sh = source_hash()
    if not force and os.environ.get("A1MPC_FORCE_BUILD") is None and library_hash() == sh:
        last_build.update(compiled=False, source_hash=sh, seconds=0.0)   # the library IS the compilation of these sources (hash compiled in), hazard-checked when it was built
        return LIB_PATH
    t_start = time.time()
    from . import isa_check
    with tempfile.TemporaryDirectory(prefix="a1mpc_build_") as tmp:  # -save-temps writes the listings into the working directory
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-save-temps", f'-DA1MPC_SOURCE_HASH="{sh}"', "-I", os.path.join(CSRC, "gfx950"),
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
    last_build.update(compiled=True, source_hash=sh, seconds=time.time() - t_start)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
