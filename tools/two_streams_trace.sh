#!/bin/bash
# rocprofv3 kernel trace of tools/two_streams.py: do k_fused dispatches of the two streams overlap in time?
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/two_streams; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/two_streams.py > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
queues = sorted({q for _, _, q in rows})
overlaps = 0
overlap_ns = 0
for i in range(1, len(rows)):
    s0, e0, q0 = rows[i - 1]
    s1, e1, q1 = rows[i]
    if q1 != q0 and s1 < e0:
        overlaps += 1
        overlap_ns += min(e0, e1) - s1
print(f"k_fused dispatches: {len(rows)} on queues {queues}")
print(f"consecutive dispatches on DIFFERENT queues that overlap in time: {overlaps}, total overlap {overlap_ns / 1e3:.1f} us")
per_q = {q: sum(1 for r in rows if r[2] == q) for q in queues}
print("dispatches per queue:", per_q)
print(open(os.path.join(root, "run.log")).read().strip().splitlines()[-1])
PY
rm -rf $OUT/trace
