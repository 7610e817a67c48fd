cd /root/repo
one() {
  local label=$1 lib=$2 e=$3
  line=$(env $e python3 bench.py --pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2 --lib $PWD/$lib --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1)
  echo "$line" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
}
for r in 1 2 3; do
one ragged prlib_amd/libprlib_hip_testhooks.so X=1
one ragged_notail tools/probe/libprlib_hip_notail.so X=1
one plain prlib_amd/libprlib_hip_testhooks.so PRL_HIP_RAGGED_UO=0
done
