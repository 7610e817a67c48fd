#!/bin/bash
# which roof bounds the round-2 stage kernels: instruction / LDS / busy counters around tools/bench_stages.py (64 A4 pages)
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/pmc_stages; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/bench_stages.py > $OUT/$tag.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "prl_hip" not in k or "k_" not in k: continue
        name = k[k.index("k_"):].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch, summed over the chip (8 XCDs, 256 CUs); valu_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GUI cycles per XCD), lds_frac = SQ_LDS_IDX_ACTIVE / (256 CUs x GUI cycles per XCD)")
for name, c in sorted(acc.items()):
    g = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else float("nan")
    gui = g("GRBM_GUI_ACTIVE") / 8
    if not gui or gui != gui or g("SQ_INSTS_VALU") < 1e6: continue
    print(f"{name:24s} launches={len(c['SQ_INSTS_VALU']):3d} gui_cycles={gui:10.4g} valu={g('SQ_INSTS_VALU'):10.4g} lds_inst={g('SQ_INSTS_LDS'):10.4g} vmem_rd={g('SQ_INSTS_VMEM_RD'):9.3g} vmem_wr={g('SQ_INSTS_VMEM_WR'):9.3g} valu_frac={g('SQ_INSTS_VALU') * 4 / (1024 * gui):5.2f} lds_frac={g('SQ_LDS_IDX_ACTIVE') / (256 * gui):5.2f}")
PY
find $OUT -name "*.csv" -size +1M -delete
