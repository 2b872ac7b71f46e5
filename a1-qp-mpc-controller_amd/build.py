"""Builds the product library liba1mpc.so (hand-written HIP for gfx950 + the C ABI of include/a1mpc.h).

hipcc cross-compiles without a GPU; the .so is written IN-TREE next to this file so it travels with the
source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).

Round 6: the library is compiled as one translation unit per (horizon, pipeline) -- csrc/a1mpc_k_*.hip -- plus
csrc/a1mpc_hip.hip (C ABI + caller-side kernels), in parallel, and linked.  Every unit's gfx950 listing goes
through the DPP hazard check (isa_check.py) and its code-object resources (VGPRs, AGPRs, spills, scratch, LDS)
are collected: a hazard, a missing hot kernel or a spill in a kernel that must not spill fails the build.
Objects are cached under build/obj by the hash of what they were compiled from, so a change of the C ABI unit
does not recompile the kernels (and vice versa).
"""
import concurrent.futures
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import tempfile
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "liba1mpc.so")
OBJ_DIR = os.path.join(_HERE, "build", "obj")          # git-ignored (build/) and gpurun-ignored: the cache stays in this container
RESOURCES_PATH = os.path.join(_HERE, "build", "kernel_resources.json")
HEADERS = ["a1mpc_common.hpp", "a1mpc_kernels.hpp", "a1mpc_solver.hpp", "a1mpc_tables.hpp", os.path.join("gfx950", "a1mpc_rowops.hpp")]
EXTENDED_HORIZONS = (14, 12, 8, 6, 4)   # the fast path's kernel family at the other even horizons (csrc/a1mpc_common.hpp, A1MPC_FAST_HORIZONS): 16-47 s per unit
KERNEL_UNITS = ([f"a1mpc_k_h{h}_{p}.hip" for h in (20, 16, 10) for p in ("split", "fused")] + [f"a1mpc_k_gen{h}_{p}.hip" for h in (20, 16, 10) for p in ("split", "fused")]
                + [f"a1mpc_k_h{h}_{p}.hip" for h in EXTENDED_HORIZONS for p in ("split", "fused")] + [f"a1mpc_k_gen{h}.hip" for h in EXTENDED_HORIZONS] + ["a1mpc_k_h1.hip"])
MAIN_UNIT = "a1mpc_hip.hip"
ID_UNIT = "a1mpc_build_id.cpp"
UNITS = KERNEL_UNITS + [MAIN_UNIT]       # (slowest first: the pool starts them in this order)
SOURCES = UNITS + [ID_UNIT] + HEADERS
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build liba1mpc.so)")


def _public_header():
    return os.path.join(os.path.dirname(_HERE), "include", "a1mpc.h")


def source_hash():
    """sha256 (16 hex digits) over the library's sources: compiled into the library (a1mpc_build_info) so that a shipped liba1mpc.so can be matched to the sources
    beside it -- file times do not survive a snapshot"""
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [_public_header()]:
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def library_hash():
    """the source hash the existing library was compiled from, or None (missing library / a build from before the hash existed).
    Read from the file's bytes (the string a1mpc_build_info() returns), NOT by loading the library: glibc matches loaded libraries by name, so a
    CDLL here would pin the old mapping and a later CDLL of the rebuilt file at the same path would silently return the pre-build code."""
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


def unit_key(unit):
    """what an object file is the compilation of: the unit, every header it can see, the flags and the compiler"""
    h = hashlib.sha256()
    deps = [unit] + (HEADERS if unit == MAIN_UNIT else HEADERS)   # (the C ABI unit includes a1mpc_common.hpp -> a1mpc_solver.hpp too, but not a1mpc_kernels.hpp)
    if unit == MAIN_UNIT:
        deps = [d for d in deps if d != "a1mpc_kernels.hpp"]
    for d in deps:
        h.update(open(os.path.join(CSRC, d), "rb").read())
    h.update(open(_public_header(), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(_compiler_id().encode())
    return h.hexdigest()[:16]


_compiler = {}


def _compiler_id():
    if "id" not in _compiler:
        try:
            _compiler["id"] = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout.strip()
        except OSError:
            _compiler["id"] = "?"
    return _compiler["id"]


def _compile_unit(unit, verbose=False):
    """one translation unit -> (object path, hazards, resources, seconds, compiled?)"""
    from . import isa_check
    stem = os.path.splitext(unit)[0]
    d = os.path.join(OBJ_DIR, f"{stem}.{unit_key(unit)}")
    obj, meta = os.path.join(d, stem + ".o"), os.path.join(d, "checked.json")
    if os.path.exists(obj) and os.path.exists(meta):
        m = json.load(open(meta))
        return obj, m["hazards"], m["resources"], m["dpp_coverage"], 0.0, False
    for old in glob.glob(os.path.join(OBJ_DIR, f"{stem}.*")):   # one cached object per unit
        shutil.rmtree(old, ignore_errors=True)
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix=f"a1mpc_{stem}_") as tmp:   # -save-temps writes the listings into the working directory
        cmd = [hipcc()] + FLAGS + ["-save-temps", "-I", os.path.join(CSRC, "gfx950"), "-I", CSRC, "-c", os.path.join(CSRC, unit), "-o", os.path.join(tmp, stem + ".o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {unit}:\n{r.stderr[-4000:]}")
        listings = glob.glob(os.path.join(tmp, f"*{ARCH}*.s"))
        if not listings:
            raise RuntimeError(f"hipcc -save-temps left no gfx950 listing to check for {unit}")
        hazards = [h for l in listings for h in isa_check.dpp_hazards(l)]
        cov, res = {}, {}
        for l in listings:
            cov.update(isa_check.dpp_coverage(l))
            res.update(isa_check.kernel_resources(l))
        os.makedirs(d, exist_ok=True)
        shutil.move(os.path.join(tmp, stem + ".o"), obj)
    json.dump({"hazards": hazards, "resources": res, "dpp_coverage": cov, "seconds": time.time() - t0}, open(meta, "w"))
    return obj, hazards, res, cov, time.time() - t0, True


last_build = {}   # what the last build() call did: {"compiled": bool, "source_hash": ..., "seconds": ..., "units": {...}} (reported by __graft_entry__.build)


def build(force=False, verbose=False, jobs=None):
    """hipcc (one process per translation unit, in parallel) -> objects -> liba1mpc.so; every unit's gfx950 listing passes the DPP hazard check and the resource
    gate of isa_check.py first: a library whose inline-asm DPP chains read a register too early after a VALU write, or whose hot kernels spill, is not linked."""
    sh = source_hash()
    if not force and os.environ.get("A1MPC_FORCE_BUILD") is None and library_hash() == sh:
        last_build.update(compiled=False, source_hash=sh, seconds=0.0, units={})   # the library IS the compilation of these sources (hash compiled in), checked when it was built
        return LIB_PATH
    if force or os.environ.get("A1MPC_FORCE_BUILD") is not None:
        shutil.rmtree(OBJ_DIR, ignore_errors=True)
    from . import isa_check
    t_start = time.time()
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = jobs or int(os.environ.get("A1MPC_BUILD_JOBS", "0")) or max(1, min(len(UNITS), os.cpu_count() or 1))
    objs, bad, resources, coverage, units = [], [], {}, {}, {}
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        futs = {u: pool.submit(_compile_unit, u, verbose) for u in UNITS}
        for u in UNITS:
            obj, hazards, res, cov, secs, compiled = futs[u].result()
            objs.append(obj); bad += hazards; resources.update(res); coverage.update(cov)
            units[u] = {"seconds": round(secs, 1), "compiled": compiled}
    bad += isa_check.coverage_gaps(None, coverage=coverage)   # the check must have seen the kernels it exists for (no fail-open)
    bad += isa_check.resource_gaps(resources)                 # ... and the hot kernels must not spill
    if bad:
        if os.path.exists(LIB_PATH):
            os.remove(LIB_PATH)
        raise RuntimeError("the generated gfx950 code fails the ISA gate (library removed):\n" + "\n".join(bad[:20]))
    with tempfile.TemporaryDirectory(prefix="a1mpc_link_") as tmp:
        ido = os.path.join(tmp, "a1mpc_build_id.o")
        subprocess.check_call([hipcc(), "-x", "c++", "-O2", "-fPIC", f'-DA1MPC_SOURCE_HASH="{sh}"', "-c", os.path.join(CSRC, ID_UNIT), "-o", ido])
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + [ido, "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    json.dump({"source_hash": sh, "kernels": resources}, open(RESOURCES_PATH, "w"), indent=1, sort_keys=True)
    last_build.update(compiled=True, source_hash=sh, seconds=time.time() - t_start, units=units, jobs=jobs)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
    print(json.dumps(last_build, indent=1))
