# config 3 / 2 A/B of two libraries on one box, interleaved: tools/dbg/ab_configs.sh libA libB [reps]
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${3:-2}); do
 for cfg in "niblack 101 0.01 2 2480 3508" "wolfjolion 101 0.01 2 2480 3508" "nick 21 -0.01 0 2480 3508" "niblack 31 0.01 0 2480 3508" "wolfjolion 31 0.01 0 2480 3508" "feng 31 0.2 0 2480 3508" "sauvola 51 0.34 0 4096 4096"; do
  set -- $cfg
  for lib in prlib_amd/libprlib_hip.so tools/ab/lib_noexit.so; do
    python3 bench.py --lib $lib --pages 128 --size $5 --height $6 --method $1 --window $2 --k $3 --morph $4 --steps 6 --warmup 2 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$1 $2', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
 done
done | tee gpurun_out/ab/configs.txt
