#!/bin/bash
# instruction counters of k_morph_bits in the morph=2 bench: tools/pmc_morph.sh <tag>
TAG=${1:-m}; OUT=$PWD/gpurun_out/pmcmorph_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 bench.py --morph 2 --steps 3 --warmup 1 --cpu-seconds 0 --check-pages 0 --traffic 0 --ceilings 0 --worst-case 0 --end-to-end 0 > $OUT/p.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "p", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_morph_bits" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(" ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc.items())))
PY
rm -rf $OUT/p
