#!/bin/bash
# A/B of the wide-window float32 threshold loop (k_fused MODE 2) against the integer loop (PRL_HIP_FLT_WIDE=0), hooks build,
# interleaved on one box: 256 A4 pages, header defaults of the three w=101 binarizers.
for rep in 1 2; do
for cfg in "niblack 101 0.01 2" "sauvola 101 0.01 2" "wolfjolion 101 0.01 2" "niblack 63 0.01 0"; do
  set -- $cfg
  for fw in 1 0; do
    PRL_HIP_FLT_WIDE=$fw python3 bench.py --hooks 1 --pages 256 --size 2480 --height 3508 --method $1 --window $2 --k $3 --morph $4 --steps 20 --warmup 3 --worst-case 0 --end-to-end 0 --traffic 0 --ceilings 0 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flt_wide=$fw', d['config']['workload'][:70], d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'call', d['roofline']['call_ms'], 'refined', d['parity']['refined_pixels'], 'bad', d['parity']['mismatching_pixels'])"
  done
done
done
