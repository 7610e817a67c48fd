#!/bin/bash
# FETCH_SIZE/WRITE_SIZE calibration run (GPU box).  Output: gpurun_out/calib/summary.txt
OUT=$PWD/gpurun_out/calib; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/calib_counters.py > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/calib_counters.py > $OUT/w.log 2>&1
{ echo "# k_calib_stream8: 4294967296 bytes read + written per dispatch, 8 B per lane"
  grep -h k_calib $(find $OUT/f $OUT/w -name "*counter_collection.csv") | awk -F, '{print $(NF-1), $NF}' ; } > $OUT/summary.txt
grep -h k_calib $(find $OUT/f $OUT/w -name "*counter_collection.csv") | head -3 >> $OUT/summary.txt
rm -rf $OUT/f $OUT/w
cat $OUT/summary.txt
