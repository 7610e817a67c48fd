#!/bin/bash
# Round 6 measurement pass (one MI355X): everything profiles/r06/ quotes.   tools/profile_r06.sh  -> gpurun_out/r06/
OUT=gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
# 1. the group kernel: text pages (1 / 8 / 64 / 256) with the phase counters, then the reference's colour scans tiled to A4
PRL_HIP_PPHT_PROF=1 timeout 300 python tools/dbg/ppht_group_prof.py 1 8 64 256 > $OUT/ppht_group_text.txt 2>&1
timeout 300 python tools/dbg/ppht_real.py 64 > $OUT/ppht_group_real.txt 2>&1
# 2. the same without the phase counters (product library), group kernel vs k_ppht_mw
timeout 600 python tools/dbg/ppht_group_check.py --pages 1 64 256 --timeout 280 > $OUT/ppht_group_check.json 2> $OUT/ppht_group_check.err
# 3. config 5 on the reference's colour scans and on synthetic scans
PRL_HIP_DEBUG=1 timeout 900 python tools/bench_real.py --chain 1 --pages 256 --check-pages 3 > $OUT/real_chain.jsonl 2> $OUT/real_chain.err
timeout 900 python tools/bench_chain5.py --pages 1024 --repeat 2 > $OUT/chain_1024.json 2> $OUT/chain_1024.err
# 4. the headline line (traffic, clock, worst case, end to end)
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# 5. config 3 / 2 and the morphology fast path A/B (same box, interleaved)
CPU_SECONDS=4 timeout 900 tools/bench_c3.sh 256 > $OUT/c3_configs_256pages.jsonl 2>&1
for rep in 1 2 3; do
  for lib in prlib_amd/libprlib_hip.so tools/ab/lib_morph_nofast.so; do
    python bench.py --lib $lib --pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 8 --warmup 2 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'call_ms': d['roofline']['call_ms'], 'kernel_ms': d['roofline']['kernel_ms']}))"
    python bench.py --lib $lib --pages 256 --size 4096 --window 31 --morph 2 --steps 8 --warmup 2 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 0 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'call_ms': d['roofline']['call_ms'], 'kernel_ms': d['roofline']['kernel_ms']}))"
  done
done > $OUT/morph_fast_path_ab.jsonl 2>&1




echo done
