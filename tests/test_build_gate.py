"""CPU: the build's resource gate (isa_check.resource_gaps) -- the hot kernels of both paths must not touch scratch memory (round 6: the general path's kernels spilled 87-754
VGPRs until then); checked on synthetic code-object notes (the gate must not fail open) and on the notes of the library built in this tree."""
import importlib
import json
import os

import __graft_entry__ as g


def _isa():
    g.load_package()
    return importlib.import_module(g.PKG_NAME + ".isa_check")


def _clean(isa):
    r = {}
    for key in isa.NO_SCRATCH:
        r["_ZN5a1mpc" + key + "EEvNS_9BatchArgsE"] = dict(vgpr=400, agpr=144, vgpr_spill=4, scratch_bytes=0, scratch_instrs=0, scratch_instrs_in_loops=0)
    for key, b, l in isa.BOUNDED_SCRATCH:
        r["_ZN5a1mpc" + key + "EEvNS_9BatchArgsEPd"] = dict(vgpr=256, agpr=0, vgpr_spill=26, scratch_bytes=b, scratch_instrs=30, scratch_instrs_in_loops=l)
    for key, v, w in isa.MAX_VGPR_BESIDE_PERSISTENT:
        r["_ZN5a1mpc18" + key + "EiPKiPi"] = dict(vgpr=v, agpr=0, vgpr_spill=0, scratch_bytes=0, scratch_instrs=0, scratch_instrs_in_loops=0)
    return r


def test_gate_accepts_agpr_parking_and_refuses_scratch():
    isa = _isa()
    r = _clean(isa)
    assert isa.resource_gaps(r) == []
    k = next(k for k in r if "a1mpc_solve_gen_kernelILi" in k)
    r[k] = dict(r[k], vgpr_spill=128, scratch_bytes=244, scratch_instrs=323, scratch_instrs_in_loops=158)   # (round 5's a1mpc_solve_gen_kernel<10, 2>)
    gaps = isa.resource_gaps(r)
    assert len(gaps) == 1 and "244 B of scratch" in gaps[0] and "158 scratch instructions inside loops" in gaps[0]


def test_gate_does_not_fail_open():
    isa = _isa()
    r = _clean(isa)
    for k in [k for k in r if "a1mpc_admm_gen_kernelILi" in k]:
        del r[k]
    assert any("saw no kernel matching a1mpc_admm_gen_kernelILi" in m for m in isa.resource_gaps(r))
    r = _clean(isa)
    k = next(k for k in r if "a1mpc_setup_gen_kernelILi20E" in k)
    r[k] = dict(r[k], scratch_instrs_in_loops=19)
    assert any("inside loops (allowed 8)" in m for m in isa.resource_gaps(r))
    r = _clean(isa)
    k = next(k for k in r if "a1mpc_order_kernel" in k)
    r[k] = dict(r[k], vgpr=24)   # (the queue-order kernel as it was until round 6: four wavefronts per SIMD x 24 registers wait for a CU without a persistent wavefront)
    assert any("no longer fit beside a persistent wavefront" in m for m in isa.resource_gaps(r))
    del r[k]
    assert any("saw no kernel matching a1mpc_order_kernel" in m for m in isa.resource_gaps(r))


def test_the_library_built_here_passes_the_gate():
    pkg = g.load_package()
    isa = _isa()
    path = pkg.build.RESOURCES_PATH
    if not os.path.exists(path):   # (a tree whose library was shipped pre-built: the gate ran where it was compiled; profiles/r06_kernel_resources.json is that build's record)
        path = os.path.join(g.ROOT, "profiles", "r06_kernel_resources.json")
    res = json.load(open(path))["kernels"]
    if not any(k.startswith("_Z") for k in res):   # the profile keeps demangled names: compare through the gate's own record instead
        assert json.load(open(path))["gate"]["violations"] == []
        return
    assert isa.resource_gaps(res) == []
    gen = {k: v for k, v in res.items() if "gen_kernel" in k or "gen_coop" in k}
    assert len(gen) >= 14 and all((v["scratch_instrs_in_loops"] or 0) <= 8 for v in gen.values())


def test_hazard_check_sees_the_extern_c_kernels(tmp_path):
    """The EKF kernel (extern "C": an unmangled label) carries inline-asm DPP chains since round 6: the DPP hazard check must parse it like the mangled kernels, find a
    hazard in it, and the coverage gate must miss it when it is gone (no fail-open)."""
    isa = _isa()
    body = lambda gap: f"""
a1mpc_ekf_kernel:                       ; @a1mpc_ekf_kernel
; %bb.0:
	v_mul_f64 v[2:3], v[4:5], v[6:7]
{gap}	v_fmac_f64_dpp v[8:9], -v[2:3], v[10:11] row_newbcast:3 row_mask:0xf bank_mask:0xf
	s_endpgm
.Lfunc_end0:
"""
    bad = tmp_path / "bad.s"; bad.write_text(body(""))
    good = tmp_path / "good.s"; good.write_text(body("\ts_nop 1\n"))
    assert len(isa.dpp_hazards(str(bad))) == 1 and "a1mpc_ekf_kernel" in isa.dpp_hazards(str(bad))[0]
    assert isa.dpp_hazards(str(good)) == []
    assert isa.dpp_coverage(str(good)) == {"a1mpc_ekf_kernel": 1}
    assert "a1mpc_ekf_kernel" in isa.EXPECTED_DPP
    gaps = isa.coverage_gaps(None, coverage={k: v for k, v in isa.EXPECTED_DPP.items() if k != "a1mpc_ekf_kernel"})
    assert any("a1mpc_ekf_kernel" in m for m in gaps), gaps
