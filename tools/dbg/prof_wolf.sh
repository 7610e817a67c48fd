#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_wolf; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --pages 256 --size 2480 --height 3508 --method wolfjolion --window 31 --k 0.01 --morph 0 --steps 5 --warmup 1 --cpu-seconds 0 --check-pages 0 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "prl_hip" in n and float(r["AverageNs"]) > 3000:
        n = n[n.index("k_"):] if "k_" in n else n
        print(f"  {n[:70]:70s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs']) / 1e6:7.3f}")
PY
