#!/bin/bash
# kernel timeline of one Wolf-Jolion step (A4 pages): what runs beside the two sweeps
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/trace_wolf
for cfg in "31 0" "101 2"; do
  set -- $cfg
  D=/tmp/trw_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --pages 256 --size 2480 --height 3508 --method wolfjolion --window $1 --k 0.01 --morph $2 --steps 6 --warmup 2 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 0 > gpurun_out/trace_wolf/w$1.log 2>&1
  t=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# find last occurrence of the first kernel of a step: use the last two k_fused<100 starts
idx=[i for i,r in enumerate(rows) if 'k_fused<100' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
prev=None
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a-3:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev)/1e3 if prev else 0
    print('%-64s start_us %8.1f dur_us %8.1f gap_us %7.1f' % (r['Kernel_Name'].replace('void prl_hip::(anonymous namespace)::','')[:64], (s-t0)/1e3, (e-s)/1e3, gap))
    prev=e
PY
  tail -1 gpurun_out/trace_wolf/w$1.log | cut -c1-200
done
