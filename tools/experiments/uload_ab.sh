#!/bin/bash
# A/B of typed UINT loads in the integer threshold loop (PRL_HIP_ULOAD=1, hooks build) on one box.
run() { PRL_HIP_ULOAD=$7 python3 bench.py --hooks 1 --pages 256 --size $5 --height $6 --method $1 --window $2 --k $3 --morph $4 --steps 20 --warmup 3 --worst-case 0 --end-to-end 0 --traffic 0 --ceilings 0 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('uload=$7', d['config']['workload'][:66], d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'call', d['roofline']['call_ms'], 'bad', d['parity']['mismatching_pixels'])"; }
for rep in 1 2; do
  for cfg in "niblack 101 0.01 2 2480 3508" "sauvola 101 0.01 2 2480 3508" "wolfjolion 101 0.01 2 2480 3508" "niblack 63 0.01 0 2480 3508" "sauvola 51 0.34 0 4096 4096"; do
    for u in 0 1; do run $cfg $u; done
  done
done
