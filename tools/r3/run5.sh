#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
timeout 900 python3 -m pytest tests/test_chain_gpu.py tests/test_deskew_gpu.py -x -q -m gpu 2>&1 | tail -5
{
echo "== adaptive (default)"; PRL_HIP_DEBUG=1 python3 tools/bench_chain5.py --pages 1024 --stages 0 --repeat 2 --check-pages 0 2> gpurun_out/r3/chain_adaptive.err | tail -1
grep "pass size\|ran .* past" gpurun_out/r3/chain_adaptive.err | tail -20
echo "== fixed 192"; PRL_HIP_CHAIN_PASS=192 python3 tools/bench_chain5.py --pages 1024 --stages 0 --repeat 2 --check-pages 0 2>/dev/null | tail -1
echo "== fixed 256"; PRL_HIP_CHAIN_PASS=256 python3 tools/bench_chain5.py --pages 1024 --stages 0 --repeat 2 --check-pages 0 2>/dev/null | tail -1
} > gpurun_out/r3/chain_sched.txt 2>&1
cat gpurun_out/r3/chain_sched.txt
