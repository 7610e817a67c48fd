#!/bin/bash
# rocprofv3 --kernel-trace --stats of the config-5 chain on 64 A4 pages (which kernels the chain consists of, and their share)
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/prof_chain; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/bench_chain5.py --pages 64 --stages 0 --check-pages 0 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/chain_64_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.05 and "prl_hip" in r["Name"]:
        n = r["Name"]; n = n[n.index("k_"):] if "k_" in n else n
        print(f"{n[:64]:64s} calls={r['Calls']:>5s} avg_ms={float(r['AverageNs']) / 1e6:9.3f} pct={r['Percentage']}")
PY
tail -1 $OUT/log.txt | cut -c1-400
