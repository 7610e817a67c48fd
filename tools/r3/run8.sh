#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
for pinned in 0 1; do PRL_HIP_DEBUG=1 python3 tools/bench_host_batch.py --pages 256 --devices 1 --pinned $pinned 2>&1 | grep -v "^\[prl_hip\]" | tail -4; done | tee gpurun_out/r3/host_batch.txt
for t in 8 16 64; do PRL_HIP_HOST_COPY_THREADS=$t python3 tools/bench_host_batch.py --pages 256 --devices 1 --pinned 0 2>&1 | tail -1 | cut -c1-260; done | tee -a gpurun_out/r3/host_batch.txt
ab() {
  for round in 1 2; do for t in 0 1; do
    line=$(PRL_HIP_C16=$t python3 bench.py --hooks 1 $@ --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 3 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c16=$t', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'mismatch', d['parity']['mismatching_pixels'])"
  done; done
}
{
echo "== A4 niblack w=101 morph=2"; ab --pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 5 --warmup 2
echo "== A4 wolf w=101 morph=2"; ab --pages 256 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --morph 2 --steps 5 --warmup 2
echo "== 4K sauvola w=101 morph=2 (API defaults)"; ab --pages 256 --window 101 --k 0.01 --morph 2 --steps 5 --warmup 2
echo "== 4K sauvola w=51"; ab --pages 256 --window 51 --steps 5 --warmup 2
echo "== 32 A4 pages niblack w=101"; ab --pages 32 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2
} > gpurun_out/r3/c16.txt 2>&1
cat gpurun_out/r3/c16.txt
timeout 900 python3 -m pytest tests/test_binarize_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/fuzz_binarize.py --seconds 120 2>&1 | tail -2
