"""CPU: the C-ABI library loads and exports every symbol include/a1mpc.h declares; no compute (there is no GPU here)
and no fallback: creating an engine without a device fails loudly."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "a1mpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(a1mpc_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(pkg):
    pkg.build.build()
    lib = C.CDLL(pkg.build.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/a1mpc.h but not exported by liba1mpc.so"
    assert set(pkg.engine.EXPORTS) <= set(names)


def test_struct_layouts_match_header(pkg):
    lib = pkg.load_library()
    cfg = pkg.Config(); lib.a1mpc_default_config(C.byref(cfg))
    assert (cfg.horizon, cfg.dt, cfg.mu, cfg.fz_max) == (10, 0.0025, 0.3, 180.0)  # S/A1Params.h:26, S/A1RobotControl.cpp:462, S/ConvexMpc.cpp:8,224
    assert (cfg.rho, cfg.sigma, cfg.alpha, cfg.eps_abs, cfg.max_iter, cfg.check_termination, cfg.scaling, cfg.warm_start) == \
        (0.1, 1e-6, 1.6, 1e-3, 4000, 25, 10, 1)
    qp = pkg.BalanceConfig(); lib.a1mpc_default_balance_config(C.byref(qp))
    assert list(qp.Q) == [1, 1, 1, 400, 400, 100] and (qp.R, qp.mu, qp.F_max) == (1e-3, 0.7, 180.0)  # S/A1RobotControl.cpp:11-15
    assert lib.a1mpc_status_string(0) == b"ok"


def test_no_cpu_fallback(pkg, scen):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = scen.scenario_T()
    cfg = pkg.make_config(sc["params"], 10)
    with pytest.raises(pkg.A1MpcError):
        pkg.Engine(cfg, 1, 0)
    bad = pkg.make_config(sc["params"], 7)
    with pytest.raises(pkg.A1MpcError):
        pkg.Engine(bad, 1, 0)


BAD_CONFIGS = [dict(p=dict(fz_min=200.0)), dict(p=dict(fz_max=-5.0, fz_min=-10.0)), dict(p=dict(mu=-0.3)), dict(p=dict(mu=float("nan"))), dict(p=dict(mass=0.0)),
               dict(p=dict(dt=0.0)), dict(p=dict(dt=float("inf"))), dict(q0=-1.0), dict(r0=-1e-6), dict(q0=float("nan")), dict(p=dict(inertia=[0.0] * 9)),
               dict(o=dict(rho=0.0)), dict(o=dict(sigma=-1e-6)), dict(o=dict(alpha=2.0)), dict(o=dict(alpha=0.0)), dict(o=dict(eps_abs=-1e-3)),
               dict(o=dict(eps_abs=0.0, eps_rel=0.0)), dict(o=dict(max_iter=0)), dict(o=dict(check_termination=-1)), dict(o=dict(scaling=-1)),
               dict(o=dict(adaptive_rho_tolerance=0.5)), dict(o=dict(adaptive_rho_interval=-25)), dict(o=dict(warm_start=3))]


def bad_config(pkg, sc, case):
    p = dict(sc["params"], **case.get("p", {}))
    if "q0" in case:
        p["q"] = [case["q0"]] + list(p["q"][1:])
    if "r0" in case:
        p["r"] = [case["r0"]] + list(p["r"][1:])
    return pkg.make_config(p, 10, **case.get("o", {}))


@pytest.mark.parametrize("case", BAD_CONFIGS, ids=[str(c) for c in BAD_CONFIGS])
def test_bad_configurations_are_refused_before_any_device_query(pkg, scen, case):
    """every configuration on which OSQP's PRIMAL / DUAL_INFEASIBLE or NON_CVX outcomes could be reached -- and every setting osqp_setup itself refuses
    (auxil.c validate_data / validate_settings) -- is A1MPC_ERR_INVALID_ARGUMENT at a1mpc_create, with or without a GPU (the check comes before hipGetDeviceCount), and
    a1mpc_last_error names the field.  include/a1mpc.h lists which OSQP statuses are unreachable as a consequence."""
    lib = pkg.load_library()
    cfg = bad_config(pkg, scen.scenario_T(), case)
    h = C.c_void_p()
    rc = lib.a1mpc_create(C.byref(cfg), 4, 0, C.byref(h))
    assert rc == 1 and not h.value, (rc, lib.a1mpc_last_error())
    assert len(lib.a1mpc_last_error()) > 10
    qp = pkg.BalanceConfig(); lib.a1mpc_default_balance_config(C.byref(qp))
    assert (qp.mu, qp.F_min, qp.F_max) == (0.7, 0.0, 180.0)


def test_the_reference_configurations_pass_validation(pkg, scen):
    """the three rosparam weight sets of the reference (config/*_a1_mpc.yaml) and the test program's (S/test/test_mpc.cpp) are accepted: without a GPU the call gets past the
    validation and fails for want of a device (3 = A1MPC_ERR_NO_DEVICE / 4 = A1MPC_ERR_HIP), with one it succeeds"""
    lib = pkg.load_library()
    for name in ("gazebo", "hardware", "isaac"):
        cfg = pkg.make_config(scen.PARAM_SETS[name] | scen.MPC_CONSTANTS, 10)
        h = C.c_void_p()
        rc = lib.a1mpc_create(C.byref(cfg), 4, 0, C.byref(h))
        assert rc in (0, 3, 4), (name, rc, lib.a1mpc_last_error())
        if rc == 0:
            lib.a1mpc_destroy(h)


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/"""
    pk = os.path.join(ROOT, "a1-qp-mpc-controller_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("oracle/a1mpc_oracle.c", "").replace("the oracle", "").lower() or f in ("scenarios.py",), (dp, f)


def test_build_hazard_check_flags_early_dpp_reads(pkg, tmp_path):
    """build() rejects a library whose inline-asm DPP chains read a VGPR fewer than two wait states after a VALU wrote it
    (isa_check.py; the failure mode is a result that depends on what the row solved before)."""
    from a1_qp_mpc_controller_amd import isa_check
    head = "_ZN5a1mpc4testEv:\n.LBB0_1:\n"
    dpp = "\tv_fmac_f64_dpp v[10:11], v[2:3], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
    cases = {
        "write_then_read": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n" + dpp, True),
        "one_between": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\tv_add_f64 v[20:21], v[6:7], v[8:9]\n" + dpp, True),
        "two_between": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\tv_add_f64 v[20:21], v[6:7], v[8:9]\n\tv_add_f64 v[22:23], v[6:7], v[8:9]\n" + dpp, False),
        "s_nop_1": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_nop 1\n" + dpp, False),
        "s_nop_0": ("\tv_cndmask_b32_e64 v3, 0, v9, s[0:1]\n\ts_nop 0\n" + dpp, True),
        "other_register": ("\tv_mul_f64 v[12:13], v[6:7], v[8:9]\n" + dpp, False),
        "own_lane_operand": ("\tv_mul_f64 v[4:5], v[6:7], v[8:9]\n" + dpp, False),  # src1 is not read through DPP
        "lds_load_is_not_valu": ("\tds_read_b64 v[2:3], v9\n" + dpp, False),
    }
    filler = "\tv_add_f64 v[30:31], v[6:7], v[8:9]\n"
    cases.update({
        # control flow: the write is the last instruction of a loop body, the DPP read the first one at the loop head
        "loop_back_edge": (dpp + filler * 3 + "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_cbranch_scc1 .LBB0_1\n", True),
        "loop_back_edge_padded": (dpp + filler * 3 + "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n" + filler * 2 + "\ts_cbranch_scc1 .LBB0_1\n", False),
        "fall_through_into_label": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n.LBB0_2:\n" + dpp, True),
        "branch_target": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_branch .LBB0_3\n.LBB0_2:\n" + filler * 3 + "\ts_endpgm\n.LBB0_3:\n" + dpp, True),
        "not_a_predecessor": ("\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_endpgm\n.LBB0_2:\n" + dpp, False),
    })
    cases.update({
        "trans_result_next": ("\tv_rcp_f64 v[2:3], v[6:7]\n\tv_fma_f64 v[12:13], -v[6:7], v[2:3], 1.0\n", True),
        "trans_result_after_one": ("\tv_rcp_f64 v[2:3], v[6:7]\n" + filler + "\tv_fma_f64 v[12:13], -v[6:7], v[2:3], 1.0\n", False),
        "trans_then_unrelated": ("\tv_rcp_f64 v[2:3], v[6:7]\n\tv_fma_f64 v[12:13], -v[6:7], v[8:9], 1.0\n", False),
    })
    swap = "\tv_permlane32_swap_b32_e32 v2, v40\n"
    cases.update({
        # twin_exchange(): v_permlane32_swap reads BOTH operands 2 wait states after a VALU write at the earliest, and writes both
        "swap_after_copy": ("\tv_mov_b32_e32 v40, v2\n" + swap, True),
        "swap_after_write_of_vdst": ("\tv_add_f64 v[2:3], v[6:7], v[8:9]\n" + filler + swap, True),
        "swap_padded": ("\tv_mov_b32_e32 v40, v2\n\ts_nop 1\n" + swap, False),
        "swap_writes_its_source": (swap + "\tv_fmac_f64_dpp v[10:11], v[40:41], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n", True),
        "swap_then_dpp_padded": (swap + filler * 2 + "\tv_fmac_f64_dpp v[10:11], v[40:41], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n", False),
    })
    for name, (body, bad) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(head + body)
        assert bool(isa_check.dpp_hazards(str(f))) == bad, name


def test_hazard_check_does_not_fail_open(pkg, tmp_path):
    """a listing in which the parser finds none of the DPP kernels must FAIL the build's check, not pass it"""
    from a1_qp_mpc_controller_amd import isa_check
    f = tmp_path / "empty.s"
    f.write_text("\t.text\n; nothing that looks like a kernel\n")
    assert isa_check.dpp_hazards(str(f)) == [] and len(isa_check.coverage_gaps([str(f)])) == len(isa_check.EXPECTED_DPP)
    g = tmp_path / "few.s"
    g.write_text("_ZN5a1mpc17a1mpc_admm_kernelILi10ELi2EEEvNS_9BatchArgsEPKdPi:\n.LBB0_1:\n\tv_fmac_f64_dpp v[10:11], v[2:3], v[4:5] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\ts_endpgm\n")
    gaps = isa_check.coverage_gaps([str(g)], {"a1mpc_admm_kernelILi10E": 1500})
    assert len(gaps) == 1 and "only 1" in gaps[0]
