#!/bin/bash
# A/B: LLVM's AMDGPU scheduling strategies for the whole library (-mllvm -amdgpu-sched-strategy=...)
cd "$(dirname "$0")/../.."
L="prlib_amd/libprlib_hip_testhooks.so tools/probe/libprlib_hip_s_max-ilp.so tools/probe/libprlib_hip_s_iterative-ilp.so tools/probe/libprlib_hip_s_max-memory-clause.so"
bash tools/r3/ab.sh "--steps 20 --warmup 3" $L
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2" $L
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2" $L
