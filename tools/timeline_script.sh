#!/bin/bash
# start/end timeline of the library's kernels in any python script: tools/timeline_script.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/tl_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "prl_hip" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
prev_end = t0
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void prl_hip::(anonymous namespace)::", "")[:28]
    print(f"{name:28s} start={(s-t0)/1e6:10.3f} ms dur={(e-s)/1e6:8.3f} ms gap_before={(s-prev_end)/1e6:8.3f} ms")
    prev_end = e
PY
rm -rf $OUT/t
