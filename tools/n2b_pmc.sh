#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and wait counters of the N2b kernel in its stand-alone harness.  usage: tools/n2b_pmc.sh OUTDIR robots
OUT=$1; N=${2:-524288}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  g=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set -d $OUT/raw_$g --output-format csv -- tools/ubench/n2b_bench $N 40 > $OUT/$g.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/raw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "contact_terrain" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for k, v in sorted(acc.items())}))   # (the later launches: every ring is full by then)
PY
grep robots $OUT/FETCH_SIZE.log | tail -1
