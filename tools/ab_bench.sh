#!/bin/bash
# A/B of library builds on one box: tools/ab_bench.sh <label>=<so-or-empty>[,ENV=VAL...] ...   (kernel ms of the headline bench)
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  so=${rest%%,*}; envs=""
  [ "$rest" != "$so" ] && envs=$(echo "${rest#*,}" | tr ',' ' ')
  for i in 1 2; do
    env $envs timeout 300 python3 bench.py ${so:+--lib $PWD/$so} --hooks $([ -z "$so" ] && echo 1 || echo 0) --traffic 0 --ceilings 0 --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', 'step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'mism', d['parity']['mismatching_pixels'], 'refined', d['parity']['refined_pixels'])"
  done
done
