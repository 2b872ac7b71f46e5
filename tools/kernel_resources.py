#!/usr/bin/env python3
"""Round 6: registers, spills, scratch and LDS of every kernel of liba1mpc.so, from the .amdgpu_metadata notes of the gfx950 listings the build keeps
(a1-qp-mpc-controller_amd/build/kernel_resources.json, written by build.py from isa_check.kernel_resources) -> profiles/r06_kernel_resources.json + a readable table.
The build itself gates on these numbers (isa_check.resource_gaps: no scratch in the hot kernels of either path).   python tools/kernel_resources.py [out.json]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
import importlib
isa_check = importlib.import_module(g.PKG_NAME + ".isa_check")
pkg.build.build()
src = json.load(open(pkg.build.RESOURCES_PATH))
rows = {}
for k, v in sorted(src["kernels"].items()):
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("a1mpc::", "").replace("void ", "")
    name = name.split("(")[0]
    rows[name] = v
out = {"source_hash": src["source_hash"], "gate": {"no_scratch": list(isa_check.NO_SCRATCH), "bounded_scratch": [list(b) for b in isa_check.BOUNDED_SCRATCH],
                                                    "violations": isa_check.resource_gaps(src["kernels"])},
       "columns": "vgpr = VGPRs + AGPRs allocated (512 = one wavefront per SIMD), agpr = of which accumulation registers, vgpr_spill = VGPRs the allocator spilled (to AGPRs where "
                  "scratch_bytes is 0, to scratch memory otherwise), scratch_instrs(_in_loops) = scratch_load / scratch_store instructions in the listing (inside a loop)",
       "kernels": rows}
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_kernel_resources.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
for n, v in rows.items():
    print("%-64s vgpr %3s agpr %3s spill %3s scratch %4s B  scratch instrs %4s (in loops %3s)" % (n[:64], v["vgpr"], v["agpr"], v["vgpr_spill"], v["scratch_bytes"], v.get("scratch_instrs"), v.get("scratch_instrs_in_loops")))
print("gate violations:", out["gate"]["violations"])
