#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
timeout 600 python3 -m pytest tests/test_binarize_gpu.py tests/test_cpp_host.py -x -q -m gpu 2>&1 | tail -3
python3 tools/bench_host_path.py | tee gpurun_out/r3/host_path.jsonl
python3 tools/bench_host_path.py --pinned | tee -a gpurun_out/r3/host_path.jsonl
