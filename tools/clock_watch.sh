#!/bin/bash
# sample sclk / power while the headline bench runs (is k_fused power-throttled?): tools/clock_watch.sh
python3 bench.py --steps 9000 --warmup 3 --cpu-seconds 0 --check-pages 0 > /tmp/bench_cw.json 2>/dev/null &
BP=$!
sleep 18
for i in 1 2 3 4 5 6; do
  timeout 20 rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.3
done
wait $BP
tail -1 /tmp/bench_cw.json | cut -c1-200
echo "idle:"; timeout 20 rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr -s ' ' | tr '\n' ';'; echo
