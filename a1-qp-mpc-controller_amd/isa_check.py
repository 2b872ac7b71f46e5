"""DPP read-after-VALU-write hazard check of a gfx950 assembly listing (hipcc -save-temps), run by build() on every build.

hipcc pads the hazards of the instructions it schedules itself, but not inside (or between) inline-asm statements, and the solver's
v_fmac_f64_dpp chains are inline asm.  Rule (gfx90a+): a VGPR written by a VALU instruction needs 2 wait states before a DPP
instruction reads it as its DPP source (src0); a VALU write of EXEC needs 5.  Every instruction in between counts as one wait
state, `s_nop N` as N + 1.  A violated hazard reads the register's previous content: results that depend on which QP the row
solved before.  The check is linear within a basic block and clears its history at a label (a hazard that spans a branch target is
not seen: the solver's blocks begin with LDS reads, not with DPP reads).
"""
import re


def _regs(tok):
    """set of VGPR indices named by an operand token (v12, v[12:13], -v[2:3], |v1|)"""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def dpp_hazards(path, key=""):
    """list of "file:line: kernel: message" strings, empty if the listing is clean"""
    out = []
    kernel = None
    hist = []  # (written VGPR set, writes_exec) of the previous instructions / wait states, newest last
    with open(path) as f:
        for ln, l in enumerate(f, 1):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                kernel = m.group(1)
                hist = []
                continue
            if re.match(r"^\.LBB\d+_\d+:", l):
                hist = []
                continue
            t = l.strip()
            if not t or t[0] in ";.#" or kernel is None or key not in kernel:
                continue
            mm = re.match(r"^([a-z][a-z0-9_]+)\s*(.*?)(\s*;.*)?$", t)
            if not mm:
                continue
            op, rest = mm.group(1), mm.group(2)
            ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)] if rest else []
            if "_dpp" in op and len(ops) >= 2:
                s0 = _regs(ops[1].split(" ")[0])
                for age, (w, ex) in enumerate(reversed(hist[-5:]), 1):  # age - 1 = wait states between the write and this read
                    if age <= 2 and (w & s0):
                        out.append(f"{path}:{ln}: {kernel[:60]}: DPP source {ops[1].split(' ')[0]} written {age} instruction(s) earlier: {t}")
                    if ex:
                        out.append(f"{path}:{ln}: {kernel[:60]}: DPP {age} wait state(s) after a VALU write of EXEC: {t}")
            if op == "s_nop":
                hist += [(set(), False)] * (int(rest.split()[0], 0) + 1)
            else:
                w = set()
                if op.startswith("v_") and ops and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    w = _regs(ops[0].split(" ")[0])
                hist.append((w, op.startswith("v_cmpx")))
            hist = hist[-8:]
    return out
