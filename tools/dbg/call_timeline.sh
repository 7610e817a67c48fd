#!/bin/bash
# Kernel timeline of ONE binarize call (which kernel runs when, on which queue), from a rocprofv3 kernel trace of bench.py.
#   tools/dbg/call_timeline.sh <method> <window> <k> <morph> [width height pages]
# The call shown is the last complete one before the profiling pass (calls are delimited by k_init_globals).
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
M=${1:-feng}; W=${2:-21}; K=${3:-0.0}; MO=${4:-2}; SZ=${5:-4096}; H=${6:-4096}; P=${7:-256}
OUT=$PWD/gpurun_out/call_timeline; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --pages $P --size $SZ --height $H --method $M --window $W --k $K --morph $MO --steps 3 --warmup 1 --cpu-seconds 0 --check-pages 0 --traffic 0 --ceilings 0 --worst-case 0 --end-to-end 0 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_init_globals" in r["Kernel_Name"]]
start, end = (idx[-3], idx[-2]) if len(idx) >= 3 else (idx[0], len(rows))
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:end]:
    n = r["Kernel_Name"]
    n = n[n.index("k_"):] if "k_" in n else n
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"  {s / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  queue {r.get('Queue_Id', '?'):>3s}  {n[:70]}")
PY
rm -rf $OUT/t
