#!/bin/bash
# A/B: edge strips in the float32 pipeline (new) against the integer loop (prev), same box, interleaved
cd "$(dirname "$0")/../.."
timeout 900 python3 -m pytest tests/test_binarize_gpu.py -m gpu -x -q 2>&1 | tail -3
N=prlib_amd/libprlib_hip_testhooks.so
O=tools/probe/libprlib_hip_prev.so
one() {  # label lib args
  local label=$1 lib=$2; shift 2
  line=$(python3 bench.py "$@" --lib $PWD/$lib --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 4 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][:60], 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for round in 1 2 3; do
  for v in prev:$O new:$N; do
    one ${v%%:*} ${v##*:} --steps 20 --warmup 3
  done
done
for round in 1 2; do
  for v in prev:$O new:$N; do
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method sauvola --window 31 --k 0.34 --morph 0 --steps 10 --warmup 2
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2
    one ${v%%:*} ${v##*:} --pages 256 --size 2480 --height 3508 --method wolfjolion --window 31 --k 0.01 --morph 0 --steps 10 --warmup 2
    one ${v%%:*} ${v##*:} --pages 1 --size 4096 --window 15 --steps 20 --warmup 3
  done
done
