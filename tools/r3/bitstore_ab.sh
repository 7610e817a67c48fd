#!/bin/bash
# A/B: the bit-plane byte store of strip_loop_f through the buffer resource (new) against a 64-bit global address (prev = in-tree hooks build of HEAD)
cd "$(dirname "$0")/../.."
N=tools/probe/libprlib_hip_b8.so
O=prlib_amd/libprlib_hip_testhooks.so
for r in 1 2; do
bash tools/r3/ab.sh "--morph 2 --steps 20 --warmup 3" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method niblack --window 31 --k 0.01 --morph 2 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph -1 --steps 10 --warmup 2" $O $N | tail -2
done
