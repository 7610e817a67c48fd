#!/bin/bash
# kernel-trace stats of one bench configuration: tools/trace_cfg.sh <tag> <bench args...>
TAG=$1; shift
OUT=$PWD/gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --check-pages 0 --traffic 0 --ceilings 0 --worst-case 0 --end-to-end 0 "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
grep -v "at::native" "$f" | head -20 > $OUT/kernel_stats_prl.csv
rm -rf $OUT/t
cat $OUT/kernel_stats_prl.csv | cut -c1-200
