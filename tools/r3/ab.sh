#!/bin/bash
# interleaved A/B of library builds on the headline bench: tools/r3/ab.sh "<bench args>" lib1 lib2 ... (two rounds)
cd "$(dirname "$0")/../.."
ARGS=$1; shift
for round in 1 2; do
  for so in "$@"; do
    line=$(python3 bench.py $ARGS --lib $PWD/$so --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'mismatch', d['parity']['mismatching_pixels'])"
  done
done
