#!/bin/bash
# round 3, GPU run 1: memory-system ceilings of the headline access shapes, rows-per-segment sweep at w=101, base bench
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
./tools/ubench/stream_ceiling 256 > gpurun_out/r3/stream_ceiling.jsonl 2> gpurun_out/r3/stream_ceiling.err
python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/r3/bench_base.json 2> gpurun_out/r3/bench_base.err
bash tools/dbg/rps_w101.sh > gpurun_out/r3/rps_w101.txt 2>&1
bash tools/bench_c3.sh 256 > gpurun_out/r3/c3_base.jsonl 2>&1
