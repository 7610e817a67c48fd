#!/bin/bash
# per-kernel times of the round-2 stages (tools/bench_stages.py, 64 A4 pages)
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/prof_stages; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/bench_stages.py > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "prl_hip" in n:
        n = n[n.index("k_"):] if "k_" in n else n
        print(f"{n[:70]:70s} calls={r['Calls']:>5s} avg_ms={float(r['AverageNs']) / 1e6:8.3f} total_ms={float(r['TotalDurationNs']) / 1e6:9.2f}")
PY
tail -1 $OUT/log.txt | cut -c1-300
