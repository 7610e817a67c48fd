#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
ab() {  # args...
  for round in 1 2; do for t in 0 1; do
    line=$(PRL_HIP_TIERS=$t python3 bench.py $@ --cpu-seconds 0 --check-pages 3 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tiers=$t', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'mismatch', d['parity']['mismatching_pixels'])"
  done; done
}
{
echo "== headline"; ab --steps 10 --warmup 3
echo "== A4 niblack w=101 morph=0"; ab --pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --steps 5 --warmup 2
echo "== A4 wolf w=101 morph=0"; ab --pages 256 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --steps 5 --warmup 2
echo "== A4 nick w=21"; ab --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --steps 5 --warmup 2
echo "== 32 pages"; ab --pages 32 --steps 20 --warmup 3
echo "== 64 pages"; ab --pages 64 --steps 20 --warmup 3
echo "== 1024 A4 w=31"; ab --pages 1024 --size 2480 --height 3508 --steps 5 --warmup 2
} > gpurun_out/r3/tiers.txt 2>&1
cat gpurun_out/r3/tiers.txt
timeout 900 python3 -m pytest tests/test_binarize_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python3 tools/fuzz_binarize.py --seconds 90 2>&1 | tail -4
