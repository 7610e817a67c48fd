#!/bin/bash
# rows per segment for the reference's header-default window (w=101) on 256 A4 pages
cd "$(dirname "$0")/../.."
for r in 0 128 192 256 384 512; do
  for m in "niblack 0.01" "wolfjolion 0.01"; do set -- $m
  PRL_HIP_ROWS_PER_SEG=$r python3 bench.py --pages 256 --size 2480 --height 3508 --method $1 --window 101 --k $2 --morph 0 --steps 5 --warmup 1 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rps=$r $1', d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['mismatching_pixels'])"
  done
done
