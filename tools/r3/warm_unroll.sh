#!/bin/bash
# warm-up loop of a segment: 2 (shipped) / 4 / 8 rows in flight
cd "$(dirname "$0")/../.."
one() {
  local label=$1 lib=$2; shift 2
  line=$(python3 bench.py "$@" --lib $PWD/$lib --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][:58], 'ms', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for round in 1 2; do
for v in u2:prlib_amd/libprlib_hip_testhooks.so u4:tools/probe/libprlib_hip_wu4.so u8:tools/probe/libprlib_hip_wu8.so; do
  l=${v%%:*}; f=${v##*:}
  one $l $f --steps 20 --warmup 3
  one $l $f --pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2
  one $l $f --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2
  one $l $f --pages 4 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 100 --warmup 10
  one $l $f --pages 1 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 100 --warmup 10
  one $l $f --pages 1 --size 4096 --window 15 --steps 100 --warmup 10
  one $l $f --pages 32 --size 4096 --window 31 --steps 20 --warmup 3
done
done
