#!/bin/bash
# Runs ON THE GPU BOX: the LDS / issue counters of the hot kernels for several library builds side by side (counters in their own runs, no trace domains).
# usage: tools/pmc_ab.sh OUTDIR "prof_target args" libA.so libB.so ...    -> OUTDIR/<lib>.json (per kernel: counter sums over the run)
OUT=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for lib in "$@"; do
  tag=$(basename $lib .so)
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
    g=$(echo $set | cut -d' ' -f1)
    A1_LIB=$lib timeout 200 rocprofv3 --pmc $set -d $OUT/raw_${tag}_$g --output-format csv -- python tools/prof_target.py $ARGS > $OUT/${tag}_$g.log 2>&1
  done
  python - $OUT $tag <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(f"{out}/raw_{tag}_*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "a1mpc" not in k or "noop" in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r.get("Dispatch_Id")))
    for k, _ in seen: calls[k] = max(calls[k], sum(1 for kk, _ in seen if kk == k))
res = {}
for k, c in acc.items():
    d = {n: v / max(1, calls[k]) for n, v in c.items()}
    if d.get("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    if d.get("SQ_WAVE_CYCLES"): d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
    d["launches"] = calls[k]; res[k] = d
json.dump(res, open(f"{out}/{tag}.json", "w"), indent=1)
print(tag, json.dumps({k: {n: (round(v, 4) if v < 10 else int(v)) for n, v in d.items()} for k, d in res.items()}))
PY
done
