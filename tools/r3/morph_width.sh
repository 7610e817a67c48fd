cd /root/repo
for sz in 4096 4000 3990 2480 2000; do
  line=$(python3 bench.py --pages 128 --size $sz --height 4096 --morph 2 --steps 10 --warmup 2 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sz', 'ms', d['ms_per_step'], 'k_fused', d['roofline']['kernel_ms'], 'rest_ms', round(d['ms_per_step']-d['roofline']['kernel_ms'],4), 'mismatch', d['parity']['mismatching_pixels'])"
done
