#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -6
timeout 400 python3 tools/fuzz_binarize.py --seconds 150 2>&1 | tail -1 > gpurun_out/r3/fuzz_binarize.txt; cat gpurun_out/r3/fuzz_binarize.txt
timeout 500 python3 tools/fuzz_stages.py --seconds 200 2>&1 | tail -3 > gpurun_out/r3/fuzz_stages.txt; cat gpurun_out/r3/fuzz_stages.txt
timeout 400 python3 tools/fuzz_chain.py --seconds 150 2>&1 | tail -2 > gpurun_out/r3/fuzz_chain.txt; cat gpurun_out/r3/fuzz_chain.txt
