#!/bin/bash
# kernel-trace stats of any python script: tools/trace_script.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
grep -v "at::native" "$f" | head -24 > $OUT/kernel_stats_prl.csv
rm -rf $OUT/t
python3 tools/print_stats.py $OUT/kernel_stats_prl.csv
