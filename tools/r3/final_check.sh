#!/bin/bash
# what the driver runs at the end of a round: GPU tests, smoke(), the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -5
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python3 bench.py ) > gpurun_out/r3/bench_final.json 2> gpurun_out/r3/bench_final.err; tail -1 gpurun_out/r3/bench_final.json | cut -c1-1500; grep real gpurun_out/r3/bench_final.err
