#!/bin/bash
# Timeline of ONE Wolf-Jolion call at the header defaults on 256 A4 pages (which kernel runs when, on which queue): the side
# stream's kernels (page-minimum border, sweep B, k_wolf_interval) should lie beside the two big sweeps.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/wolf_timeline; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --pages ${PAGES:-256} --size 2480 --height 3508 --method wolfjolion --window ${WIN:-101} --k 0.01 --morph ${MORPH:-2} --steps 3 --warmup 1 --cpu-seconds 0 --check-pages 0 --traffic 0 --ceilings 0 --worst-case 0 --end-to-end 0 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete call: find the last k_init / first sweep of the last step = last occurrence of kernel names containing 'k_fusedILi100'
idx = [i for i, r in enumerate(rows) if "k_fused<100" in r["Kernel_Name"] or "k_fusedILi100" in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 else idx[-1]      # (the profiling pass after the timed steps is the last; take the one before)
end = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[max(0, start - 3):end]:
    n = r["Kernel_Name"]
    n = n[n.index("k_"):] if "k_" in n else n
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"  {s / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  queue {r.get('Queue_Id', '?'):>3s}  {n[:60]}")
PY
