#!/bin/bash
# A/B: ragged uo on (default) / off (PRL_HIP_RAGGED_UO=0), hooks build, same box, interleaved
cd "$(dirname "$0")/../.."
timeout 900 python3 -m pytest tests/test_binarize_gpu.py -m gpu -x -q 2>&1 | tail -3
one() {  # label env args
  local label=$1 e=$2; shift 2
  line=$(env $e python3 bench.py "$@" --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 4 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][:64], 'Mpx/s', round(d['value']), 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for round in 1 2 3; do
  for v in off:PRL_HIP_RAGGED_UO=0 on:PRL_HIP_RAGGED_UO=1; do
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2
  done
done
python3 tools/fuzz_binarize.py --seconds 240 --seed 4242 2>&1 | tail -1
