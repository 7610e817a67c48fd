#!/bin/bash
# config 2 (one 4096x4096 page on the device, w=15): rows per segment below the present floor of 16
cd "$(dirname "$0")/../.."
for round in 1 2; do
for r in 0 16 12 8 6 4; do
  line=$(env PRL_HIP_ROWS_PER_SEG=$r python3 bench.py --pages 1 --size 4096 --window 15 --steps 200 --warmup 20 --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rps $r', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
done
for r in 0 16 8; do
  line=$(env PRL_HIP_ROWS_PER_SEG=$r python3 bench.py --pages 1 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 200 --warmup 20 --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('A4 niblack101 rps $r', 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
done
done
