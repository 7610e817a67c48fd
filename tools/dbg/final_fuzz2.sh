mkdir -p gpurun_out/fuzz_final2
timeout 900 python tools/fuzz_chain.py --seconds 600 --seed 6401 > gpurun_out/fuzz_final2/chain_6401.json 2>&1 &
timeout 900 python tools/fuzz_stages.py --seconds 600 --seed 6402 --only deskew,houghp --real 0.5 --max-side 1200 > gpurun_out/fuzz_final2/deskew_6402.json 2>&1 &
PRL_HIP_FUSED_QINT=1 timeout 900 python tools/fuzz_binarize.py --hooks 1 --seconds 600 --seed 6403 --wide 0.9 --real 0.3 > gpurun_out/fuzz_final2/binarize_qint1_6403.json 2>&1 &
timeout 900 python tools/fuzz_binarize.py --seconds 600 --seed 6404 --wide 0.5 --real 0.5 --adversarial 0.1 > gpurun_out/fuzz_final2/binarize_6404.json 2>&1 &
wait
for f in gpurun_out/fuzz_final2/*.json; do echo $f; tail -c 500 $f; echo; done
