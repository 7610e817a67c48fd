#!/bin/bash
# sclk / power samples (with wall-clock stamps) while the config-5 chain runs twice; the library prints stamped pass times
cd "$(dirname "$0")/../.."
PRL_HIP_DEBUG=1 python3 tools/bench_chain5.py --pages 1024 --stages 0 --check-pages 0 --repeat 2 > /tmp/cc.out 2>&1 &
BP=$!
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  echo "$(date +%s.%N | cut -c1-14) $(timeout 20 rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr -s ' ' | tr '\n' ';')"
  sleep 0.2
done
wait $BP
grep "prl chain" /tmp/cc.out
tail -1 /tmp/cc.out | cut -c1-120
