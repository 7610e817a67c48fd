#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
{
echo "== headline"; bash tools/r3/ab.sh "--steps 10 --warmup 3" tools/probe_build/libprlib_r2.so prlib_amd/libprlib_hip.so
echo "== A4 niblack w=31"; bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method niblack --window 31 --k 0.01 --steps 5 --warmup 2" tools/probe_build/libprlib_r2.so prlib_amd/libprlib_hip.so
echo "== 32 pages (strong-scaling proxy)"; bash tools/r3/ab.sh "--pages 32 --steps 20 --warmup 3" tools/probe_build/libprlib_r2.so prlib_amd/libprlib_hip.so
} > gpurun_out/r3/ab_lospec.txt 2>&1
cat gpurun_out/r3/ab_lospec.txt
timeout 900 python3 -m pytest tests/test_binarize_gpu.py tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -5
