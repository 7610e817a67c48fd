#!/bin/bash
# Speed-of-light probes of k_fused: the shipped library and the three PRL_PROBE builds on the same box,
# same session, headline workload.  Output: gpurun_out/probe_fused.jsonl (one bench line per build, tagged).
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/probe_fused.jsonl
: > $out
for tag in ${PROBE_TAGS:-shipped probe1 probe2 probe3 probe4 probe5 shipped_again}; do
  so=prlib_amd/libprlib_hip.so
  case $tag in probe*) so=tools/probe_build/libprlib_$tag.so;; esac
  line=$(python3 bench.py --lib $PWD/$so --traffic 0 --ceilings 0 --steps 10 --warmup 3 --cpu-seconds 0 --check-pages 0 2>/dev/null | tail -1)
  echo "{\"build\": \"$tag\", \"line\": $line}" >> $out
done
python - <<'PY'
import json
for l in open('gpurun_out/probe_fused.jsonl'):
    d = json.loads(l); r = d['line']['roofline']
    print(d['build'], 'ms_per_step', d['line']['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'])
PY
