#!/bin/bash
# Runs ON THE GPU BOX: issue / LDS counters of the EKF kernel for several builds of tools/ubench/ekf_bench side by side (counters in their own runs, no trace domains).
# usage: tools/ekf_pmc.sh OUTDIR robots bin[:ENV=VAL] ...   -> OUTDIR/ekf_pmc.txt
OUT=$1; N=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for spec in "$@"; do
  bin=${spec%%:*}; envs=""; [ "$spec" != "$bin" ] && envs=${spec#*:}
  tag=$(basename $bin)_${envs//=/}
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
    g=$(echo $set | cut -d' ' -f1)
    env $envs timeout 200 rocprofv3 --pmc $set -d $OUT/raw_${tag}_$g --output-format csv -- $bin $N 4 > $OUT/${tag}_$g.log 2>&1
  done
  python - $OUT $tag <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(f"{out}/raw_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "ekf" not in k or "init" in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in acc.items():
    d = {m: v / n[k][m] for m, v in c.items()}
    print(tag, k, json.dumps({m: int(v) for m, v in sorted(d.items())}))
PY
done > $OUT/ekf_pmc.txt 2>&1
cat $OUT/ekf_pmc.txt
