#!/bin/bash
# per-iteration slope of the ADMM kernel for several library builds on one box: kernel ms at fixed 1 / 101 iterations, 16384 QPs (8 per resident row)
# usage: tools/slope_probe.sh libA.so libB.so ...   -> us per QP-iteration = (ms(101) - ms(1)) / 100 / 8 * 1000
cd "$GRAFT_REPO_ROOT"; timeout 300 python -c "import torch" # page the image in before the timed children
for k in 1 101; do timeout 500 python tools/ab_probe.py "$@" --only 16384 --fixed $k > /tmp/slope_$k.json 2>/tmp/slope_err_$k.txt || cat /tmp/slope_err_$k.txt; done
python - "$@" <<'PY'
import json, os, sys
a = json.load(open("/tmp/slope_1.json")); b = json.load(open("/tmp/slope_101.json"))
for lib in sys.argv[1:]:
    k = os.path.basename(lib)
    print("%-24s 1 it: %.3f ms   101 it: %.3f ms   slope %.3f us per QP-iteration (history) / %.3f (index)" % (
        k, a[k]["16384"]["history_ms"], b[k]["16384"]["history_ms"], (b[k]["16384"]["history_ms"] - a[k]["16384"]["history_ms"]) * 1000 / 100 / 8,
        (b[k]["16384"]["index_ms"] - a[k]["16384"]["index_ms"]) * 1000 / 100 / 8))
PY
