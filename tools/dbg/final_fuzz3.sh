mkdir -p gpurun_out/fuzz_final3
timeout 1500 python tools/fuzz_binarize.py --seconds 1200 --seed 6501 --wide 0.7 --real 0.4 --adversarial 0.05 > gpurun_out/fuzz_final3/binarize_6501.json 2>&1 &
timeout 1500 python tools/fuzz_binarize.py --seconds 1200 --seed 6502 --wide 1.0 --methods 0,1,2,3 --real 0.2 > gpurun_out/fuzz_final3/binarize_wide_6502.json 2>&1 &
PRL_HIP_FUSED_QINT=0 timeout 1500 python tools/fuzz_binarize.py --hooks 1 --seconds 1200 --seed 6503 --wide 0.7 > gpurun_out/fuzz_final3/binarize_qint0_6503.json 2>&1 &
timeout 1500 python tools/fuzz_stages.py --seconds 1200 --seed 6504 --real 0.5 --max-side 1000 > gpurun_out/fuzz_final3/stages_6504.json 2>&1 &
timeout 1500 python tools/fuzz_chain.py --seconds 1200 --seed 6505 > gpurun_out/fuzz_final3/chain_6505.json 2>&1 &
wait
for f in gpurun_out/fuzz_final3/*.json; do echo $f; tail -c 500 $f; echo; done
