#!/bin/bash
# chain config 5, 1024 A4 pages: overlap on/off, k_ppht priority on/off (same box, same session)
cd "$(dirname "$0")/../.."
for cfg in "1 0" "0 0" "1 3"; do
  set -- $cfg
  echo "== overlap=$1 prio=$2"
  PRL_HIP_DEBUG=1 PRL_HIP_CHAIN_OVERLAP=$1 PRL_HIP_PPHT_PRIO=$2 timeout 300 python tools/bench_chain5.py --pages 1024 --stages 0 --check-pages 0 --repeat 2 2>&1 | grep -v amdgpu.ids | tail -22
done
