#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/lvprof; rm -rf $OUT; mkdir -p $OUT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/dbg/lv_prof.py > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_lv", "k_bg", "k_warp")):
        print(n[:90].replace("prl_hip::(anonymous namespace)::",""), r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
rm -rf $OUT/trace
