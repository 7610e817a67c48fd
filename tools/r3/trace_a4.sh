#!/bin/bash
# kernel trace of two A4 configurations: where do the 0.2-0.3 ms beside k_fused go?
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/trace_a4
for cfg in "nick 21 -0.01" "niblack 31 0.01" "sauvola 31 0.34"; do
  set -- $cfg
  D=/tmp/tr_$1_$2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --pages 256 --size 2480 --height 3508 --method $1 --window $2 --k $3 --morph 0 --steps 10 --warmup 2 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 0 > gpurun_out/trace_a4/$1_$2.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/trace_a4/$1_$2_kernel_stats.csv
  t=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 - $t > gpurun_out/trace_a4/$1_$2_timeline.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 12 dispatches: name, duration, gap from previous end
prev=None
for r in rows[-14:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev)/1e3 if prev else 0
    print('%-60s dur_us %8.1f gap_us %8.1f' % (r['Kernel_Name'][:60], (e-s)/1e3, gap))
    prev=e
PY
  tail -1 gpurun_out/trace_a4/$1_$2.log | cut -c1-300
  cat gpurun_out/trace_a4/$1_$2_timeline.txt
done
