#!/bin/bash
# kernels of one single-page call (config 2: one 4096^2 page, Sauvola w=15): names, durations, gaps
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/c2_trace; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 bench.py --pages 1 --size 4096 --window 15 --steps 20 --warmup 3 --cpu-seconds 0 --check-pages 0 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(os.path.join(sys.argv[1], "t", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# the last 3 steps
prl = [i for i, r in enumerate(rows) if "k_fused" in r[2]]
start = prl[-3] - 2
prev = None
for s, e, n in rows[start:]:
    print(f"{(s - rows[start][0]) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {((s - prev) / 1e3 if prev else 0):6.1f}  {n}")
    prev = e
PY
tail -1 $OUT/log.txt | cut -c1-200
