mkdir -p gpurun_out/qint
timeout 600 python -m pytest tests/test_binarize_gpu.py -m gpu -q -k "wolf or wide_windows" 2>&1 | tail -2
PRL_HIP_FUSED_QINT=2 timeout 300 python tools/fuzz_binarize.py --hooks 1 --seconds 100 --seed 91 --wide 0.8 --methods 2 --real 0.3 > gpurun_out/qint/fuzz_wolf.json 2>&1; tail -c 330 gpurun_out/qint/fuzz_wolf.json
for rep in 1 2 3 4; do
 for q in 1 2; do
    PRL_HIP_FUSED_QINT=$q python3 bench.py --hooks 1 --pages 256 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --morph 2 --steps 6 --warmup 2 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('qint=$q', 'wolfjolion 101 morph 2 A4', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['call_ms'], d['parity']['mismatching_pixels'])"
 done
done | tee gpurun_out/qint/ab_wolf_a.txt
