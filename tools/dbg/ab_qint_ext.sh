mkdir -p gpurun_out/qint
timeout 900 python -m pytest tests/test_binarize_gpu.py tests/test_fullsize_gpu.py tests/test_batchsize_gpu.py -m gpu -q -x 2>&1 | tail -2
PRL_HIP_FUSED_QINT=2 timeout 300 python tools/fuzz_binarize.py --hooks 1 --seconds 120 --seed 93 --wide 1.0 --methods 0,1,2,3 > gpurun_out/qint/fuzz_ext.json 2>&1; tail -c 330 gpurun_out/qint/fuzz_ext.json
for rep in 1 2 3 4; do
 for cfg in "niblack 101 0.01 2 2480 3508" "sauvola 101 0.34 0 2480 3508" "sauvola 51 0.34 0 4096 4096"; do
  set -- $cfg
  for q in 0 1 2; do
    PRL_HIP_FUSED_QINT=$q python3 bench.py --hooks 1 --pages 256 --size $5 --height $6 --method $1 --window $2 --k $3 --morph $4 --steps 6 --warmup 2 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('qint=$q', '$1 $2 morph $4 $5x$6', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel'), d['parity']['mismatching_pixels'])"
  done
 done
done | tee gpurun_out/qint/ab_ext.txt
