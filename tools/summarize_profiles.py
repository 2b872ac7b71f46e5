#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of gpurun_out/prof_final/ into profiles/ (tracked)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof_" + TAG); dst = os.path.join(ROOT, "profiles")
def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


f = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(dst, TAG + "_kernel_stats_bench_batch4096_h10.csv"))
f = newest(os.path.join(src, "trace", "**", "*domain_stats.csv"))
if f:
    shutil.copy(f, os.path.join(dst, TAG + "_domain_stats.csv"))
log = os.path.join(src, "bench_under_rocprof.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, TAG + "_bench_under_rocprof.json"), "w").write(lines[-1])
f = newest(os.path.join(src, "general_trace", "**", "*kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(dst, TAG + "_kernel_stats_general_path.csv"))
f = newest(os.path.join(src, "trace_depth1", "**", "*kernel_stats.csv"))
if f:
    shutil.copy(f, os.path.join(dst, TAG + "_kernel_stats_bench_depth1_batch4096_h10.csv"))
log = os.path.join(src, "bench_depth1_under_rocprof.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, TAG + "_bench_depth1_under_rocprof.json"), "w").write(lines[-1])


def overlap_summary(trace_dir):
    """from the kernel trace's timestamps: how the launches of the PIPELINED timed region (the first 2 + 10 ADMM launches of the process: warm-up, timed) sit on the time axis --
    span per launch (the rate launches complete at) against the mean duration of one launch's kernels (longer than the span when launches overlap)"""
    f = newest(os.path.join(trace_dir, "**", "*kernel_trace.csv"))
    if not f:
        return None
    rows = [r for r in csv.DictReader(open(f)) if "a1mpc" in r.get("Kernel_Name", "") and "noop" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    admm = [r for r in rows if "a1mpc_admm" in r["Kernel_Name"]]
    setup = [r for r in rows if "setup_kernel" in r["Kernel_Name"]]
    if len(admm) < 12 or len(setup) < 12:
        return None
    timed_a, timed_s = admm[2:12], setup[2:12]
    t0 = min(int(r["Start_Timestamp"]) for r in timed_s); t1 = max(int(r["End_Timestamp"]) for r in timed_a)
    dur = lambda rs: float(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)) / len(rs) * 1e-6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed_a)
    return {"timed_launches": 10, "span_ms_per_launch": (t1 - t0) * 1e-6 / 10, "mean_admm_kernel_ms": dur(timed_a), "mean_setup_kernel_ms": dur(timed_s),
            "admm_kernels_in_flight_mean": busy / float(t1 - t0), "streams": sorted({r.get("Stream_Id", r.get("Queue_Id", "?")) for r in timed_a})}


ov = {"depth2_default": overlap_summary(os.path.join(src, "trace")), "depth1": overlap_summary(os.path.join(src, "trace_depth1"))}
if any(ov.values()):
    json.dump(ov, open(os.path.join(dst, TAG + "_kernel_trace_overlap.json"), "w"), indent=1)
    print(json.dumps(ov, indent=1))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
pmc_files = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    d = f.split(os.sep)[-3]
    if d not in pmc_files or os.path.getmtime(f) > os.path.getmtime(pmc_files[d]):
        pmc_files[d] = f
for f in pmc_files.values():
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "a1mpc" in k:
            if "noop" in k:
                continue  # the launch-latency filler of the batch-1 probe, not part of a solve
            short = ("setup_kernel" if "setup" in k else "admm_kernel" if "admm" in k else "order_kernel" if "order" in k
                     else "solve_kernel(fused)" if "solve" in k else k.split("(")[0].split("::")[-1])
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not pmc_files:
    print("no PMC passes in", src, "-- profiles/%s_pmc_summary.json left as it is" % TAG)
    sys.exit(0)
summ = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in d.items()} for k, d in acc.items()}
# HBM bytes per launch.  MI355X_MICROARCH.md (HBM / rocprofv3): on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -- double it -- and says to calibrate on a
# known byte count in one's own access pattern.  Calibration on this pipeline's own known counts (4096 QPs, h = 10): the set-up kernel WRITES the hand-off records,
# 4096 x Prep<10>::STRIDE doubles = 21.3 MB, and WRITE_SIZE reports 21.3 MB (writes: as counted); the ADMM kernel READS those records once (+ 0.3 MB of inputs) and
# FETCH_SIZE reports 11.6 MB, the set-up kernel reads 5.5 MB of inputs and FETCH_SIZE reports 3.2 MB: both a factor 2.0 / 1.75 low -- the guide's factor applies.
fetch = sum(summ.get(k, {}).get("FETCH_SIZE", {}).get("mean_per_launch", 0.0) for k in summ)
write = sum(summ.get(k, {}).get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) for k in summ)
summ["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
summ["_hbm_bytes_per_solve_batch_launch"] = summ["hbm_bytes_per_launch"]
summ["_hbm_bytes_as_counted"] = {"FETCH_SIZE_bytes": fetch * 1024.0, "WRITE_SIZE_bytes": write * 1024.0, "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950: 128-byte read requests tallied at 64 B; "
                                 "checked against the hand-off record's known size, see tools/summarize_profiles.py)"}
# executed FP64: wave-level instruction counts x the live lanes (set-up kernel: 4 rows x 12 lanes of 64; ADMM kernel: 2 QPs x (main + twin row) x 12 --
# the twin row's share of the work that it computes redundantly with its main row, the costate / roll-out recurrences, is counted: it is executed)
live = {"setup_kernel": 48, "admm_kernel": 48}
ex = 0.0
for k, lanes in live.items():
    c = summ.get(k, {})
    g_ = lambda n: c.get(n, {}).get("mean_per_launch", 0.0)
    ex += lanes * (2.0 * g_("SQ_INSTS_VALU_FMA_F64") + g_("SQ_INSTS_VALU_ADD_F64") + g_("SQ_INSTS_VALU_MUL_F64"))
summ["executed_fp64_flops_per_launch"] = ex
_c = summ.get("admm_kernel", {}); _g = lambda n: _c.get(n, {}).get("mean_per_launch", 0.0)
summ["executed_fp64_flops_admm_kernel"] = 48 * (2.0 * _g("SQ_INSTS_VALU_FMA_F64") + _g("SQ_INSTS_VALU_ADD_F64") + _g("SQ_INSTS_VALU_MUL_F64"))   # the persistent ADMM kernel alone (bench.py: admm_kernel_executed_frac)
# the other shapes (collect_profiles.sh: A1_SHAPE passes of tools/prof_target.py): per kernel and launch, counters + kernel durations
by_cfg, shapes = {}, {}
for d in sorted(glob.glob(os.path.join(src, "shape_*_SQ_INSTS_VALU_FMA_F64"))) + sorted(glob.glob(os.path.join(src, "shape_*_SQ_LDS_BANK_CONFLICT"))):
    shape = os.path.basename(d).split("_")[1]   # "8192x16", or "8192x16cu" = the same batch on the CU-wide kernel (collect_profiles.sh)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        a2 = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "a1mpc" in k and "noop" not in k:
                short = "setup_kernel" if "setup" in k else "admm_kernel" if "admm" in k else "order_kernel" if "order" in k else k.split("(")[0].split("::")[-1]
                a2[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in a2.items():
            shapes.setdefault(shape, {}).setdefault(k, {}).update({c: sum(v) / len(v) for c, v in dd.items()})
for shape, ks in shapes.items():
    h_ = int("".join(ch for ch in shape.split("x")[1] if ch.isdigit()))
    # ADMM kernel: two QPs per wavefront at h = 10, one (a main / twin pair of rows) from h = 16 on.  The executed-FP64 figure of 8192 x h16 is taken from the pass on the
    # ONE-WAVE twin-pair kernels (A1MPC_CU_WIDE=0 A1MPC_QUAD=0: 24 live lanes in every instruction, exact); the CU-wide kernel ("..cu": five QPs on four wavefronts)
    # executes the same arithmetic per QP bit for bit
    # "..q": the quad-of-rows kernel as it runs (h = 20: 48 live lanes, rows 1 / 3 repeating the sweeps of rows 0 / 2) -- what the FP64 pipe issues, repeats included
    # "..cu" (h = 16, the CU-wide kernel as it runs since the quads: wave 0 two twin pairs, waves 1-3 a quad each -- 48 live lanes in every instruction): issued, repeats included
    lanes = {"setup_kernel": 48, "admm_kernel": 48 if (h_ == 10 or shape.endswith("q") or shape.endswith("cu")) else 24}
    e_ = 0.0
    for k, ln_ in lanes.items():
        c = ks.get(k, {})
        e_ += ln_ * (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0))
    by_cfg[shape] = e_
    for k, c in ks.items():
        if c.get("SQ_LDS_IDX_ACTIVE"): c["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_WAVE_CYCLES"): c["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    f = newest(os.path.join(src, "shape_%s_trace" % shape, "**", "*kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, TAG + "_kernel_stats_first_solves_%s.csv" % shape.replace("x", "_h")))
summ["executed_fp64_flops_per_launch_by_config"] = by_cfg
summ["other_shapes_per_launch"] = shapes
for k in ("admm_kernel", "setup_kernel"):
    c = summ.get(k, {})
    g_ = lambda n: c.get(n, {}).get("mean_per_launch", 0.0)
    if g_("SQ_LDS_IDX_ACTIVE"): summ.setdefault("_derived", {})[k + "_lds_conflict_frac"] = g_("SQ_LDS_BANK_CONFLICT") / g_("SQ_LDS_IDX_ACTIVE")
    if g_("SQ_WAVE_CYCLES"): summ.setdefault("_derived", {})[k + "_wait_any_frac"] = g_("SQ_WAIT_ANY") / g_("SQ_WAVE_CYCLES")
summ["_note"] = ("rocprofv3 --pmc, one counter group per pass (tools/collect_profiles.sh), 4096 QPs h=10 default OSQP settings cold start "
                 "(tools/prof_target.py); FETCH_SIZE / WRITE_SIZE in KiB per kernel launch; one solve_batch = setup_kernel + admm_kernel. "
                 "hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE: the guide's gfx950 correction of FETCH_SIZE, confirmed on the known size of the set-up -> ADMM hand-off records.")
json.dump(summ, open(os.path.join(dst, TAG + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)[:3000])
