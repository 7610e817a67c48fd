#!/bin/bash
# SQ / LDS counter passes of any python script: tools/pmc_script.sh <tag> <script> [args...]
# (counters in their own runs, no tracing domains beside them)
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
    --output-format csv -d $OUT/pmc_sq -- python3 "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD \
    --output-format csv -d $OUT/pmc_sq2 -- python3 "$@" > $OUT/pmc_sq2.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for sub in ("pmc_sq", "pmc_sq2"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "prl_hip" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        short = k.replace("void prl_hip::(anonymous namespace)::", "")[:40]
        for c, v in sorted(d.items()):
            print(f"{short:40s} {c:24s} n={len(v):4d} avg={sum(v)/len(v):.6g}")
PY
rm -rf $OUT/pmc_sq $OUT/pmc_sq2
