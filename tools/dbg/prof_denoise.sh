#!/bin/bash
# per-kernel times of prl::denoise (8 x 4096^2 x 3): tools/dbg/prof_denoise.sh
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/prof_denoise; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/bench_denoise.py > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(r["Name"][:90], r["Calls"], "avg_ms=%.3f" % (float(r["AverageNs"]) / 1e6), "pct=" + r["Percentage"])
PY
