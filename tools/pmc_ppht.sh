#!/bin/bash
# HBM-side traffic of k_ppht (deskew of 256 A4 pages): tools/pmc_ppht.sh   (separate --pmc passes, as the guide prescribes)
OUT=$PWD/gpurun_out/pmc_ppht; mkdir -p $OUT; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/dbg/deskew_sweep.py 256 > $OUT/$tag.log 2>&1
  echo "rc=$? $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ppht" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"k_ppht {c} n={len(v)} values={[f'{x:.4g}' for x in v]}")
PY
find $OUT -name "*.csv" -size +1M -delete
