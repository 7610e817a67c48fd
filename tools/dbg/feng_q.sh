mkdir -p gpurun_out/qint
timeout 900 python -m pytest tests/test_binarize_gpu.py -m gpu -q 2>&1 | tail -2
PRL_HIP_FUSED_QINT=2 timeout 300 python tools/fuzz_binarize.py --hooks 1 --seconds 100 --seed 95 --wide 1.0 --methods 4 --real 0.3 > gpurun_out/qint/fuzz_feng.json 2>&1; tail -c 330 gpurun_out/qint/fuzz_feng.json
for rep in 1 2 3; do
 for q in 0 2; do
    PRL_HIP_FUSED_QINT=$q python3 bench.py --hooks 1 --pages 256 --size 2480 --height 3508 --method feng --window 51 --k 0.2 --morph 2 --steps 6 --warmup 2 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('qint=$q', 'feng 51 morph 2 A4', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['parity']['mismatching_pixels'])"
 done
done | tee gpurun_out/qint/ab_feng.txt
