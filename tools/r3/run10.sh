#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
{
echo "== 32 x 4K pages, rows per segment forced (hooks build) vs chosen (0)"
for r in 1 2; do for rps in 64 0 58 52 48 72 40; do
  PRL_HIP_ROWS_PER_SEG=$rps python3 bench.py --hooks 1 --pages 32 --steps 30 --warmup 5 --traffic 0 --ceilings 0 --cpu-seconds 0 --check-pages 1 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rps=$rps', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'mismatch', d['parity']['mismatching_pixels'])"
done; done
} > gpurun_out/r3/strong_proxy.txt 2>&1
cat gpurun_out/r3/strong_proxy.txt
python3 tools/dbg/soak.py 2>&1 | tail -1
