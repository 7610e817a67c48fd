#!/bin/bash
# A/B of the pipelined morphology (closing of chunk i beside the sweeps of chunk i + 1): chunks 1 (off) / 2 / 4 / 8, hooks build,
# interleaved on one box; 256 A4 pages at the header defaults of the three w = 101 binarizers, and 256 4K pages Sauvola w = 31 morph 2.
run() {  # method window k morph size height chunks
  PRL_HIP_MORPH_CHUNKS=$7 python3 bench.py --hooks 1 --pages 256 --size $5 --height $6 --method $1 --window $2 --k $3 --morph $4 --steps 20 --warmup 3 --worst-case 0 --end-to-end 0 --traffic 0 --ceilings 0 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunks=$7', d['config']['workload'][:66], d['value'], 'ms', d['ms_per_step'], 'call', d['roofline']['call_ms'], 'bad', d['parity']['mismatching_pixels'])"
}
for rep in 1 2; do
  for cfg in "niblack 101 0.01 2 2480 3508" "sauvola 101 0.01 2 2480 3508" "wolfjolion 101 0.01 2 2480 3508" "sauvola 31 0.34 2 4096 4096"; do
    for ch in 1 2 4 8; do run $cfg $ch; done
  done
done
