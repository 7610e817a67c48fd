#!/bin/bash
# instruction / LDS counters of the NL-means kernels (prl::denoise on 8 x 4096^2 x 3): tools/pmc_nlm.sh > profiles/rNN/pmc_nlm.txt
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/pmc_nlm; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/bench_denoise.py --steps 1 > $OUT/$tag.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_nlm" in k:
            name = k[k.index("k_nlm"):].split("(")[0]
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:24s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.5g}")
PY
find $OUT -name "*.csv" -size +1M -delete
