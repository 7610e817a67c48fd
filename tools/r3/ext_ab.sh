#!/bin/bash
# A/B: extended last strip on (default) / off (PRL_HIP_EXT_STRIP=0), hooks build, same box, interleaved
cd "$(dirname "$0")/../.."
one() {  # label env args
  local label=$1 e=$2; shift 2
  line=$(env $e python3 bench.py "$@" --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 4 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][:64], 'Mpx/s', round(d['value']), 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for round in 1 2; do
  for v in off:PRL_HIP_EXT_STRIP=0 on:PRL_HIP_EXT_STRIP=1; do
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method sauvola --window 101 --k 0.01 --morph 0 --steps 10 --warmup 2
    one ${v%%:*} ${v##*:} --pages 64 --size 4096 --method sauvola --window 101 --k 0.34 --morph 0 --steps 10 --warmup 2
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method sauvola --window 51 --k 0.34 --morph 0 --steps 10 --warmup 2
  done
done
python3 tools/fuzz_binarize.py --seconds 240 --seed 777 2>&1 | tail -1
