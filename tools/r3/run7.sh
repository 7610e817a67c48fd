#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -8
for pinned in 0 1; do python3 tools/bench_host_batch.py --pages 256 --devices 1 --pinned $pinned 2>&1 | tail -1; done | tee gpurun_out/r3/host_batch.jsonl
python3 tools/dbg/soak.py 2>&1 | tail -1
