#!/bin/bash
# A/B of library builds on the config-5 chain (same box): tools/dbg/chain_ab.sh <so> [<so> ...]
cd "$(dirname "$0")/../.."
for so in "$@"; do
  echo "== $so"
  PRLIB_HIP_SO=$so PRL_HIP_DEBUG=1 timeout 300 python tools/bench_chain5.py --pages 1024 --stages 0 --check-pages 0 --repeat 2 2>&1 | grep -v amdgpu.ids | tail -10
  PRLIB_HIP_SO=$so timeout 200 python tools/dbg/deskew_sweep.py 2>&1 | tail -4
done
