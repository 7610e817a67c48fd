#!/bin/bash
# BASELINE config 3 sweep (A4@300dpi pages, Niblack / Wolf-Jolion / NICK) + config 2; one JSON line each, the CPU oracle on
# all host cores timed beside every configuration (cpu_baseline; CPU_SECONDS=0 skips it).  hbm_frac_whole_call = algorithmic bytes /
# the WHOLE call (every sweep of Wolf-Jolion, refinement, morphology: prl_hip_last_call_ms); hbm_frac_dominant_kernel = / the threshold sweep alone.
P=${1:-64}
for cfg in "niblack 101 0.01 2" "wolfjolion 101 0.01 2" "nick 21 -0.01 0" "niblack 31 0.01 0" "wolfjolion 31 0.01 0" "nick 31 -0.01 0"; do
  set -- $cfg
  python3 bench.py --pages $P --size 2480 --height 3508 --method $1 --window $2 --k $3 --morph $4 --steps 5 --warmup 1 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds ${CPU_SECONDS:-6} 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'workload': d['config']['workload'], 'Mpixels/s': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'call_ms': d['roofline']['call_ms'], 'hbm_frac_whole_call': d['roofline']['frac_whole_call'], 'hbm_frac_dominant_kernel': d['roofline']['frac_dominant_kernel'], 'cpu_baseline': d['cpu_baseline'], 'parity': d['parity']}))"
done
python3 bench.py --pages 1 --size 4096 --window 15 --steps 20 --warmup 3 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'workload': d['config']['workload'], 'Mpixels/s': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel_ms': d['roofline']['kernel_ms'], 'call_ms': d['roofline']['call_ms'], 'hbm_frac_whole_call': d['roofline']['frac_whole_call'], 'hbm_frac_dominant_kernel': d['roofline']['frac_dominant_kernel'], 'parity': d['parity']}))"
