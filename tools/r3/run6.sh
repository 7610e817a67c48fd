#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
( time python3 bench.py ) > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
tail -1 gpurun_out/r3/bench_default.json; grep real gpurun_out/r3/bench_default.err
( time timeout 1500 python3 -m pytest tests/test_batchsize_gpu.py -x -q -m gpu ) 2>&1 | tail -8
bash tools/bench_c3.sh 256 > gpurun_out/r3/c3_256.jsonl 2>&1; cat gpurun_out/r3/c3_256.jsonl | cut -c1-400
python3 tools/bench_denoise.py --pages 64 > gpurun_out/r3/c4_denoise_64.json 2>&1; tail -1 gpurun_out/r3/c4_denoise_64.json
python3 tools/bench_chain5.py --pages 1024 --stage-pages 64 --repeat 2 > gpurun_out/r3/chain_1024.json 2> gpurun_out/r3/chain_1024.err; tail -1 gpurun_out/r3/chain_1024.json
