#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
for pinned in 0 1; do PRL_HIP_DEBUG=1 python3 tools/bench_host_batch.py --pages 256 --devices 1 --pinned $pinned 2>&1 | grep -v "^\[prl_hip\]" | tail -2; done | tee gpurun_out/r3/host_batch.txt
{
echo "== 32 x 4K pages: round-2 library, this build (interleaved, 3 rounds)"
for r in 1 2 3; do for so in tools/probe_build/libprlib_r2.so prlib_amd/libprlib_hip.so; do
  python3 bench.py --lib $PWD/$so --pages 32 --steps 30 --warmup 5 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
done; done
echo "== 64 x 4K pages"
for r in 1 2; do for so in tools/probe_build/libprlib_r2.so prlib_amd/libprlib_hip.so; do
  python3 bench.py --lib $PWD/$so --pages 64 --steps 30 --warmup 5 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
done; done
echo "== 256 x 4K pages (headline)"
for r in 1 2 3; do for so in tools/probe_build/libprlib_r2.so prlib_amd/libprlib_hip.so; do
  python3 bench.py --lib $PWD/$so --steps 20 --warmup 3 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
done; done
} > gpurun_out/r3/strong_proxy.txt 2>&1
cat gpurun_out/r3/strong_proxy.txt
timeout 600 python3 -m pytest tests/test_binarize_gpu.py -x -q -m gpu 2>&1 | tail -2
