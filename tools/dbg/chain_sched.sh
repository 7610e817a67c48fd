#!/bin/bash
# pass schedules of the config-5 chain (first pass / main pass), 1024 A4 pages, same box
cd "$(dirname "$0")/../.."
for cfg in "380 380" "64 256" "64 380" "32 256" "128 256" "64 192" "64 320"; do
  set -- $cfg
  echo -n "first=$1 main=$2: "
  PRL_HIP_CHAIN_FIRST_PASS=$1 PRL_HIP_CHAIN_PASS=$2 timeout 300 python tools/bench_chain5.py --pages 1024 --stages 0 --check-pages 0 --repeat 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chain_one_call_s'], d['pages_per_s'])"
done
