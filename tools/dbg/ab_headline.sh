# headline A/B on one box, interleaved: tools/dbg/ab_headline.sh libA libB [reps]
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${3:-4}); do
  for lib in $1 $2; do
    python bench.py --lib $lib --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done | tee gpurun_out/ab/headline.txt
