cd /root/repo
one() {  # label env args
  local label=$1; shift
  line=$(env "$@" python3 bench.py --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2 --hooks 1 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for r in 1 2; do
one off_tiers PRL_HIP_RAGGED_UO=0
one on_tiers PRL_HIP_RAGGED_UO=1
one off_rps128 PRL_HIP_RAGGED_UO=0 PRL_HIP_TIERS=0 PRL_HIP_ROWS_PER_SEG=128
one on_rps128 PRL_HIP_RAGGED_UO=1 PRL_HIP_TIERS=0 PRL_HIP_ROWS_PER_SEG=128
one off_rps256 PRL_HIP_RAGGED_UO=0 PRL_HIP_TIERS=0 PRL_HIP_ROWS_PER_SEG=256
one on_rps256 PRL_HIP_RAGGED_UO=1 PRL_HIP_TIERS=0 PRL_HIP_ROWS_PER_SEG=256
done
