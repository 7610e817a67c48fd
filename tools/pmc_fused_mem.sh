#!/bin/bash
# memory-pipeline counters of k_fused (typed-load float pipeline vs integer pipeline): tools/pmc_fused_mem.sh <tag>
# every profiler run sits under its own timeout: an unknown counter name can hang rocprofv3
TAG=${1:-flt}; OUT=$PWD/gpurun_out/pmcmem_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --check-pages 0 --traffic 0 --ceilings 0 --worst-case 0 --end-to-end 0"
run() { n=$1; shift; timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 bench.py $ARGS > $OUT/$n.log 2>&1; }
run p1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_ANY
run p3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run p2 TA_BUSY_avr TD_BUSY_avr
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for sub in ("p1", "p3", "p2"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fused" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print(f"{c:28s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
    if not acc:
        print(sub, "no data:", open(os.path.join(root, sub + ".log")).read()[-300:].replace("\n", " | "))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
