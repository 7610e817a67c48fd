#!/bin/bash
# round 3, GPU run 2: pure-arithmetic probes of k_fused at the occupancies an on-chip ring would leave; C3 per-kernel times
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
out=gpurun_out/r3/probe6.txt; : > $out
run() {  # tag so lds
  line=$(PRL_PROBE_LDS=$3 python3 bench.py --lib $PWD/$2 --traffic 0 --ceilings 0 --steps 10 --warmup 3 --cpu-seconds 0 --check-pages 0 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 lds=$3', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" >> $out
}
run shipped prlib_amd/libprlib_hip.so 0
for lds in 0 10240 13600 20000; do run probe6 tools/probe_build/libprlib_probe6.so $lds; done
for lds in 0 10240 13600; do run probe4 tools/probe_build/libprlib_probe4.so $lds; done
run probe3 tools/probe_build/libprlib_probe3.so 0
run shipped_again prlib_amd/libprlib_hip.so 0
cat $out
bash tools/dbg/prof_c3.sh > gpurun_out/r3/prof_c3.txt 2>&1
cat gpurun_out/r3/prof_c3.txt
