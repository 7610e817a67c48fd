#!/bin/bash
# Wolf-Jolion: longest tier 512 (as for the others) / 256 / 128 / 64 rows - sweep B lasts as long as its longest segment
cd "$(dirname "$0")/../.."
one() {  # label env args
  local label=$1 e=$2; shift 2
  line=$(env $e python3 bench.py "$@" --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][15:64], 'Mpx/s', round(d['value']), 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for round in 1 2; do
  for t in 512 256 128 64; do
    one tmax$t PRL_HIP_WOLF_TIER_MAX=$t --pages 256 --size 2480 --height 3508 --method wolfjolion --window 31 --k 0.01 --morph 0 --steps 10 --warmup 2
    one tmax$t PRL_HIP_WOLF_TIER_MAX=$t --pages 256 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2
    one tmax$t PRL_HIP_WOLF_TIER_MAX=$t --pages 64 --size 4096 --method wolfjolion --window 31 --k 0.3 --morph 0 --steps 10 --warmup 2
  done
done
