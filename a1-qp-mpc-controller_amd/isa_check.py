"""DPP read-after-VALU-write hazard check of a gfx950 assembly listing (hipcc -save-temps), run by build() on every build.

hipcc pads the hazards of the instructions it schedules itself, but not inside (or between) inline-asm statements, and the solver's
v_fmac_f64_dpp chains are inline asm.  Rule (gfx90a+): a VGPR written by a VALU instruction needs 2 wait states before a DPP
instruction reads it as its DPP source (src0), and the same before v_permlane16/32_swap reads either operand; a VALU write of EXEC needs 5; the result of a transcendental instruction
(v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) needs 1 before another VALU instruction reads it.  Every instruction in
between counts as one wait state, `s_nop N` as N + 1.  A violated hazard reads the register's previous content: results that depend on which QP the row
solved before.

Control flow: the listing is split into basic blocks at labels; a block starts from the merged tails of ALL its predecessors (the
fall-through one and every block that branches to its label, loop back-edges included), merged per age, so a write at the end of a
loop body is seen by a DPP read at the loop head.
"""
import re

_DEPTH = 6  # wait states of history that matter (EXEC rule: 5)


def _regs(tok):
    """set of VGPR indices named by an operand token (v12, v[12:13], -v[2:3], |v1|)"""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def _parse(path, key):
    """kernels -> list of blocks {label, instrs: [(line_no, text, op, ops)], targets: [labels], falls: bool}"""
    kernels = {}
    kernel = None
    cur = None
    with open(path) as f:
        for ln, l in enumerate(f, 1):
            m = re.match(r"^(_Z\w+|a1mpc_\w+):", l)   # (mangled names, and the extern "C" kernels of a1mpc_hip.hip: the EKF kernel has DPP chains of its own since round 6)
            if m:
                kernel = m.group(1) if key in m.group(1) else None
                if kernel:
                    cur = dict(label=None, instrs=[], targets=[], falls=True)
                    kernels[kernel] = [cur]
                continue
            if kernel is None:
                continue
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = dict(label=m.group(1), instrs=[], targets=[], falls=True)
                kernels[kernel].append(cur)
                continue
            t = l.strip()
            if not t or t[0] in ";.#":
                if t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"):
                    kernel = None
                continue
            mm = re.match(r"^([a-z][a-z0-9_]+)\s*(.*?)(\s*;.*)?$", t)
            if not mm:
                continue
            op, rest = mm.group(1), mm.group(2)
            ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)] if rest else []
            cur["instrs"].append((ln, t, op, ops))
            b = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
            if b:
                cur["targets"].append(b.group(1))
            cur["falls"] = not (op in ("s_branch", "s_endpgm", "s_setpc_b64"))
    return kernels


_TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def _step(hist, op, ops, rest_first):
    """history after one instruction (list of (written VGPRs, writes EXEC, is transcendental), newest last)"""
    if op == "s_nop":
        hist = hist + [(frozenset(), False, False)] * (int(rest_first, 0) + 1)
    else:
        w = frozenset()
        if op.startswith("v_") and ops and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            w = frozenset(_regs(ops[0].split(" ")[0]))
            if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap_")) and len(ops) >= 2:  # both operands are written
                w = w | frozenset(_regs(ops[1].split(" ")[0]))
        hist = hist + [(w, op.startswith("v_cmpx"), op.startswith(_TRANS))]
    return hist[-_DEPTH:]


def _merge(tails):
    """per-age union of several history tails (aligned at the newest entry)"""
    out = []
    for age in range(1, _DEPTH + 1):
        w = set()
        ex = tr = False
        for t in tails:
            if len(t) >= age:
                w |= t[-age][0]
                ex = ex or t[-age][1]
                tr = tr or t[-age][2]  # conservative: "some predecessor's write at this age was transcendental"
        out.append((frozenset(w), ex, tr))
    return list(reversed(out))


def dpp_hazards(path, key=""):
    """list of "file:line: kernel: message" strings, empty if the listing is clean"""
    out = []
    for kernel, blocks in _parse(path, key).items():
        # pass 1: the tail every block leaves behind when entered with an empty history (a block of >= _DEPTH wait states forgets its entry)
        tails = []
        for b in blocks:
            h = []
            for _, _, op, ops in b["instrs"]:
                h = _step(h, op, ops, ops[0].split()[0] if ops else "0")
            tails.append(h)
        by_label = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
        preds = {i: [] for i in range(len(blocks))}
        for i, b in enumerate(blocks):
            if b["falls"] and i + 1 < len(blocks):
                preds[i + 1].append(i)
            for t in b["targets"]:
                if t in by_label:
                    preds[by_label[t]].append(i)
        # short blocks pass their own entry history on: iterate the entry histories to a fixed point (bounded: the history is _DEPTH deep)
        entry = [[] for _ in blocks]
        for _ in range(_DEPTH + 1):
            changed = False
            for i, b in enumerate(blocks):
                exits = []
                for p in preds[i]:
                    h = list(entry[p])
                    for _, _, op, ops in blocks[p]["instrs"]:
                        h = _step(h, op, ops, ops[0].split()[0] if ops else "0")
                    exits.append(h)
                new = _merge(exits) if exits else []
                if new != entry[i]:
                    entry[i] = new
                    changed = True
            if not changed:
                break
        # pass 2: check
        for i, b in enumerate(blocks):
            h = list(entry[i])
            for ln, t, op, ops in b["instrs"]:
                if op.startswith("v_") and not op.startswith(_TRANS) and h and h[-1][2]:
                    srcs = set().union(*[_regs(o.split(" ")[0]) for o in (ops if op.startswith("v_fmac") else ops[1:])]) if ops else set()
                    if h[-1][0] & srcs:
                        out.append(f"{path}:{ln}: {kernel[:60]}: result of a transcendental instruction read by the next instruction: {t}")
                if op.startswith(("v_permlane16_swap", "v_permlane32_swap")) and len(ops) >= 2:  # twin_exchange: both operands are read, 2 wait states after a VALU write
                    rd = _regs(ops[0].split(" ")[0]) | _regs(ops[1].split(" ")[0])
                    for age, (w, _ex, _tr) in enumerate(reversed(h[-2:]), 1):
                        if w & rd:
                            out.append(f"{path}:{ln}: {kernel[:60]}: v_permlane*_swap operand written {age} instruction(s) earlier: {t}")
                if "_dpp" in op and len(ops) >= 2:
                    s0 = _regs(ops[1].split(" ")[0])
                    for age, (w, ex, _tr) in enumerate(reversed(h[-5:]), 1):  # age - 1 = wait states between the write and this read
                        if age <= 2 and (w & s0):
                            out.append(f"{path}:{ln}: {kernel[:60]}: DPP source {ops[1].split(' ')[0]} written {age} instruction(s) earlier: {t}")
                        if ex:
                            out.append(f"{path}:{ln}: {kernel[:60]}: DPP {age} wait state(s) after a VALU write of EXEC: {t}")
                h = _step(h, op, ops, ops[0].split()[0] if ops else "0")
    return out


# The check must not fail OPEN: if a change of listing format, name mangling or block labels made _parse() find nothing, dpp_hazards() would
# return [] and every build would pass.  build() therefore also demands that the kernels made of inline-asm DPP chains were found and that a
# plausible number of v_fmac_f64_dpp instructions was inspected in each.
EXPECTED_DPP = {"a1mpc_admm_gen_cu_kernelILi10E": 1500, "a1mpc_admm_kernelILi10E": 1500, "a1mpc_admm_kernelILi16E": 2500, "a1mpc_admm_kernelILi20E": 3000, "a1mpc_admm_cu_kernelILi16E": 2500, "a1mpc_setup_kernelILi10E": 60,
                "a1mpc_solve_kernelILi10E": 1500, "a1mpc_solve_coop_kernelILi10E": 1500, "a1mpc_solve_gen_kernelILi10E": 1500,
                "a1mpc_ekf_kernel": 1200}   # (the EKF: 1288 = 2 x 378 in the elimination + 28 x 19 in the rank update)


def dpp_coverage(path):
    """{kernel: number of v_fmac_f64_dpp instructions the hazard check looked at}"""
    return {k: sum(1 for b in blocks for (_, _, op, _) in b["instrs"] if "_dpp" in op and op.startswith("v_fmac_f64")) for k, blocks in _parse(path, "").items()}


def coverage_gaps(paths, expected=None, coverage=None):
    """list of messages: expected kernels that were not parsed, or parsed with too few DPP instructions (`coverage`: the merged dpp_coverage() of every unit's listing)"""
    expected = EXPECTED_DPP if expected is None else expected
    cov = dict(coverage or {})
    for p in paths or []:
        cov.update(dpp_coverage(p))
    out = []
    for key, least in expected.items():
        hits = [n for k, n in cov.items() if key in k]
        if not hits:
            out.append(f"hazard check saw no kernel matching {key} (listing format changed?)")
        elif min(hits) < least:
            out.append(f"hazard check inspected only {min(hits)} v_fmac_f64_dpp in {key} (expected >= {least})")
    return out


# ---- code-object resources (round 6): registers, spills, scratch and LDS of every kernel, from the .amdgpu_metadata notes of the listing ----------------
def kernel_resources(path):
    """{mangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_static_bytes, scratch_instrs, scratch_instrs_in_loops}} of one gfx950 listing"""
    import yaml
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"\.amdgpu_metadata\n(.*?)\n\s*\.end_amdgpu_metadata", txt, re.S):
        doc = yaml.safe_load(m.group(1).replace("---", "", 1).rsplit("...", 1)[0])
        for k in (doc or {}).get("amdhsa.kernels", []):
            out[k[".name"]] = {"vgpr": k.get(".vgpr_count"), "agpr": k.get(".agpr_count"), "sgpr": k.get(".sgpr_count"), "vgpr_spill": k.get(".vgpr_spill_count", 0),
                               "sgpr_spill": k.get(".sgpr_spill_count", 0), "scratch_bytes": k.get(".private_segment_fixed_size", 0), "lds_static_bytes": k.get(".group_segment_fixed_size", 0),
                               "max_flat_workgroup_size": k.get(".max_flat_workgroup_size")}
    # scratch traffic, and how much of it sits inside a loop (a block that a later block branches back to)
    for kernel, blocks in _parse(path, "").items():
        if kernel not in out:
            continue
        by_label = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
        in_loop = [False] * len(blocks)
        for i, b in enumerate(blocks):
            for t in b["targets"]:
                j = by_label.get(t)
                if j is not None and j <= i:
                    for k in range(j, i + 1):
                        in_loop[k] = True
        tot = loop = 0
        for i, b in enumerate(blocks):
            n = sum(1 for (_, _, op, _) in b["instrs"] if op.startswith("scratch_"))
            tot += n
            loop += n if in_loop[i] else 0
        out[kernel]["scratch_instrs"] = tot
        out[kernel]["scratch_instrs_in_loops"] = loop
    return out


# Kernels that must not touch scratch memory (a VGPR spilled to scratch in one of these is a performance bug the build refuses; `vgpr_spill_count` > 0 with 0 B of
# scratch is a value parked in the AGPR half of the register file, which costs a v_accvgpr move and is accepted): the kernels the BASELINE configs run on the fast
# path -- and, round 6, every kernel of the general path.  The A/B fallback instantiations behind environment switches (one-wave h = 16 kernels, the non-quad h = 20 kernel)
# and the update-path / contact-schedule variants at h = 16 / 20 are reported in kernel_resources.json but not gated.
NO_SCRATCH = ("a1mpc_admm_kernelILi10ELi2E", "a1mpc_admm_kernelILi20ELi1ELb0ELb1ELb0ELb1E", "a1mpc_admm_cu_kernelILi16ELb0ELb1ELb0ELb1E", "a1mpc_setup_kernelILi10E",
              "a1mpc_setup_kernelILi16ELi1ELb0E", "a1mpc_setup_kernelILi20ELi1ELb0E", "a1mpc_solve_kernelILi10E", "a1mpc_solve_kernelILi16E", "a1mpc_solve_kernelILi20E",
              "a1mpc_solve_coop_kernelILi",
              # round 6: the general path (per-step feet / contact schedules) -- until then 87-754 spilled VGPRs per kernel, 145-503 scratch instructions inside loops
              "a1mpc_solve_gen_kernelILi", "a1mpc_solve_gen_coop_kernelILi", "a1mpc_admm_gen_kernelILi", "a1mpc_admm_gen_cu_kernelILi", "a1mpc_setup_gen_kernelILi10E"
              # ... and every kernel of the extended horizons (4, 6, 8, 12, 14: the fast path's family as it instantiates there -- a horizon whose kernels would spill is not offered)
              ) + tuple(f"a1mpc_{k}_kernelILi{h}E" for h in (4, 6, 8, 12, 14) for k in ("admm", "setup", "solve")) + tuple(f"a1mpc_setup_gen_kernelILi{h}E" for h in (4, 6, 8, 12))
# ... and kernels that may park a few long-lived values (pointers, the rotation) in scratch ACROSS their loops but not inside them: (pattern, scratch bytes, scratch instructions in loops)
BOUNDED_SCRATCH = (("a1mpc_setup_gen_kernelILi14E", 64, 0), ("a1mpc_setup_gen_kernelILi16E", 64, 0), ("a1mpc_setup_gen_kernelILi20E", 128, 8))   # (two wavefronts per SIMD: 256 registers; a dozen long-lived values wait in scratch while the Ruiz passes run)


# ... and kernels that have to fit BESIDE a resident persistent wavefront (424 of a SIMD's 512 registers are taken, 88 are left): (pattern, registers per wavefront, wavefronts
# per SIMD of one workgroup).  The queue-order kernel is one 1024-thread workgroup = four wavefronts per SIMD: at 24 registers it had to wait for a CU without any persistent
# wavefront (0.4-0.5 ms behind a batch in flight, profiles/r06_setup_ahead.md); at 8 it starts at once
MAX_VGPR_BESIDE_PERSISTENT = (("a1mpc_order_kernel", 16, 4),)


def resource_gaps(resources, no_scratch=None):
    """list of messages: a kernel of NO_SCRATCH with scratch memory, or no kernel at all for one of its patterns (no fail-open)"""
    no_scratch = NO_SCRATCH if no_scratch is None else no_scratch
    out = []
    for key in no_scratch:
        hits = {k: v for k, v in resources.items() if key in k}
        if not hits:
            out.append(f"resource gate saw no kernel matching {key} (listing format changed?)")
        for k, v in hits.items():
            if (v.get("scratch_bytes") or 0) > 0:
                out.append(f"{k[:90]}: {v.get('vgpr_spill')} spilled VGPRs, {v.get('scratch_bytes')} B of scratch per lane ({v.get('scratch_instrs_in_loops')} scratch instructions inside loops)")
    if no_scratch is NO_SCRATCH:
        for key, max_vgpr, waves in MAX_VGPR_BESIDE_PERSISTENT:
            hits = {k: v for k, v in resources.items() if key in k}
            if not hits:
                out.append(f"resource gate saw no kernel matching {key} (listing format changed?)")
            for k, v in hits.items():
                if (v.get("vgpr") or 0) > max_vgpr:
                    out.append(f"{k[:90]}: {v.get('vgpr')} registers per wavefront x {waves} wavefronts per SIMD no longer fit beside a persistent wavefront (allowed {max_vgpr})")
        for key, max_bytes, max_in_loops in BOUNDED_SCRATCH:
            hits = {k: v for k, v in resources.items() if key in k}
            if not hits:
                out.append(f"resource gate saw no kernel matching {key} (listing format changed?)")
            for k, v in hits.items():
                if (v.get("scratch_bytes") or 0) > max_bytes or (v.get("scratch_instrs_in_loops") or 0) > max_in_loops:
                    out.append(f"{k[:90]}: {v.get('scratch_bytes')} B of scratch per lane (allowed {max_bytes}), {v.get('scratch_instrs_in_loops')} scratch instructions inside loops (allowed {max_in_loops})")
    return out
