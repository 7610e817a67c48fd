#!/bin/bash
# small batches: the automatic segment length (floor 8 rows) against forced 16 rows (the floor before)
cd "$(dirname "$0")/../.."
one() {
  local label=$1 e=$2; shift 2
  line=$(env $e python3 bench.py "$@" --steps 100 --warmup 10 --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][:58], 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for round in 1 2; do
for n in 1 2 4; do
  one auto X=1 --pages $n --size 4096 --window 31
  one rps16 PRL_HIP_ROWS_PER_SEG=16 --pages $n --size 4096 --window 31
done
for n in 1 2 4; do
  one auto X=1 --pages $n --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2
  one rps32 PRL_HIP_ROWS_PER_SEG=32 --pages $n --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2
done
one auto X=1 --pages 1 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --morph 2
one rps32 PRL_HIP_ROWS_PER_SEG=32 --pages 1 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --morph 2
one auto X=1 --pages 1 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0
one rps16 PRL_HIP_ROWS_PER_SEG=16 --pages 1 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0
done
