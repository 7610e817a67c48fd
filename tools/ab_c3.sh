#!/bin/bash
# rows-per-segment sweep on a C3 configuration: tools/ab_c3.sh <method> <window> <k> <morph> rps...
M=$1; W=$2; K=$3; MO=$4; shift 4
for r in "$@"; do
  for i in 1 2; do
    PRL_HIP_ROWS_PER_SEG=$r timeout 200 python3 bench.py --pages 64 --size 2480 --height 3508 --method $M --window $W --k $K --morph $MO --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$M w$W rps$r', 'step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'mism', d['parity']['mismatching_pixels'])"
  done
done
