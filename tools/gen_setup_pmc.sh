#!/bin/bash
# Runs ON THE GPU BOX: issue / wait counters of the general path's kernels (set-up | ADMM) under tools/general_path_stage_probe.py.  usage: tools/gen_setup_pmc.sh OUTDIR
OUT=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  g=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set -d $OUT/raw_$g --output-format csv -- python tools/general_path_stage_probe.py > $OUT/$g.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/raw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if "setup" not in k: continue
        acc[(k, r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), c in sorted(acc.items()):
    d = {m: int(sum(v) / len(v)) for m, v in sorted(c.items())}
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc: d["wait_any_frac"] = round(d.get("SQ_WAIT_ANY", 0) / wc, 3); d["wait_inst_frac"] = round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3); d["active_frac"] = round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    print(k, "grid", g, json.dumps(d))
PY
