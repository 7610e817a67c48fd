#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
ab() {
  for round in 1 2 3; do for t in 0 1; do
    line=$(PRL_HIP_PVRING=$t python3 bench.py --hooks 1 $@ --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 4 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pvring=$t', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'mismatch', d['parity']['mismatching_pixels'])"
  done; done
}
{
echo "== headline"; ab --steps 10 --warmup 3
echo "== A4 nick w=21"; ab --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --steps 5 --warmup 2
echo "== A4 sauvola w=15 morph=2"; ab --pages 256 --size 2480 --height 3508 --window 15 --morph 2 --steps 5 --warmup 2
echo "== 32 pages"; ab --pages 32 --steps 20 --warmup 3
} > gpurun_out/r3/pvring.txt 2>&1
cat gpurun_out/r3/pvring.txt
