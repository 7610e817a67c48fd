#!/bin/bash
# kernel times of a C3 configuration (256 A4 pages, Niblack w=101 morph=2) and of the 4K morph=2 bench
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for cfg in "--pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2" "--pages 256 --size 4096 --morph 2" "--pages 256 --size 2480 --height 3508 --method niblack --window 31 --k 0.01 --morph 2"; do
  OUT=$PWD/gpurun_out/prof_c3; rm -rf $OUT; mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py $cfg --steps 5 --warmup 1 --cpu-seconds 0 --check-pages 0 > $OUT/log.txt 2>&1
  f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "prl_hip" in n and float(r["AverageNs"]) > 20000:
        n = n[n.index("k_"):] if "k_" in n else n
        print(f"  {n[:60]:60s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs']) / 1e6:7.3f}")
PY
done
