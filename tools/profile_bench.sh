#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the headline bench.
# Usage: tools/profile_bench.sh <tag> [extra bench.py args...]
# Results land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --cpu-seconds 0 --check-pages 0 --worst-case 0 --end-to-end 0 $*"
# 1. kernel trace + stats (no counters in this run)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --traffic 0 --ceilings 0 $ARGS > "$OUT/trace.log" 2>&1
# 2. PMC passes, each in its own run (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --traffic 0 --ceilings 0 $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --traffic 0 --ceilings 0 $ARGS > "$OUT/pmc_write.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
    --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --traffic 0 --ceilings 0 $ARGS > "$OUT/pmc_sq.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py --traffic 0 --ceilings 0 $ARGS > "$OUT/pmc_sq2.log" 2>&1
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
# keep what is judged (stats + summary), drop the bulky per-dispatch traces so gpurun_out stays < 64 MiB
mkdir -p "$OUT/keep"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/keep/kernel_stats.csv" \;
for d in pmc_fetch pmc_write pmc_sq pmc_sq2; do
  f=$(find "$OUT/$d" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|k_fused|k_threshold|k_nlm" "$f" | head -400 > "$OUT/keep/${d}_counters_head.csv"
done
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_sq2"
cat "$OUT/summary.txt"
