mkdir -p gpurun_out/fuzz_final
timeout 600 python tools/fuzz_binarize.py --seconds 420 --seed 6301 --wide 0.6 --real 0.3 --adversarial 0.05 > gpurun_out/fuzz_final/binarize_6301.json 2>&1 &
timeout 600 python tools/fuzz_binarize.py --seconds 420 --seed 6302 --wide 0.9 --methods 0,1,2,3 > gpurun_out/fuzz_final/binarize_6302.json 2>&1 &
timeout 600 python tools/fuzz_stages.py --seconds 420 --seed 6303 --real 0.5 --max-side 900 > gpurun_out/fuzz_final/stages_6303.json 2>&1 &
wait
tail -c 400 gpurun_out/fuzz_final/binarize_6301.json; tail -c 400 gpurun_out/fuzz_final/binarize_6302.json; tail -c 600 gpurun_out/fuzz_final/stages_6303.json
timeout 600 python -m pytest tests/test_binarize_gpu.py -m gpu -q -k "wide_windows_float_rows" 2>&1 | tail -2
