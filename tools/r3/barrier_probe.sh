#!/bin/bash
# What does one s_barrier per output row cost k_fused when its wavefronts run in blocks of 4 / 8?  (probe build: -DPRL_PROBE_BARRIER)
cd "$(dirname "$0")/../.."
run() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  line=$(env "$@" python3 bench.py --steps 20 --warmup 3 --lib $PWD/$lib --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
H=prlib_amd/libprlib_hip_testhooks.so
B=tools/probe/libprlib_hip_barrier.so
for round in 1 2; do
  run default_wpb1_tiers $H X=1
  run wpb1_rps315 $H PRL_HIP_ROWS_PER_SEG=315 PRL_HIP_TIERS=0
  run wpb8_rps315 $H PRL_HIP_WPB=8 PRL_HIP_ROWS_PER_SEG=315 PRL_HIP_TIERS=0
  run wpb8_rps315_barrier $B PRL_HIP_WPB=8 PRL_HIP_ROWS_PER_SEG=315 PRL_HIP_TIERS=0
  run wpb4_rps315 $H PRL_HIP_WPB=4 PRL_HIP_ROWS_PER_SEG=315 PRL_HIP_TIERS=0
  run wpb4_rps315_barrier $B PRL_HIP_WPB=4 PRL_HIP_ROWS_PER_SEG=315 PRL_HIP_TIERS=0
done
