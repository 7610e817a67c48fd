#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
timeout 800 python3 tools/fuzz_binarize.py --seconds 600 --seed 31 2>&1 | tail -1 > gpurun_out/r3/fuzz_binarize_long.txt; cat gpurun_out/r3/fuzz_binarize_long.txt
timeout 600 python3 tools/fuzz_chain.py --seconds 400 --seed 32 2>&1 | tail -1 > gpurun_out/r3/fuzz_chain_long.txt; cat gpurun_out/r3/fuzz_chain_long.txt
timeout 600 python3 tools/fuzz_stages.py --seconds 400 --seed 33 2>&1 | tail -1 > gpurun_out/r3/fuzz_stages_long.txt; cat gpurun_out/r3/fuzz_stages_long.txt
