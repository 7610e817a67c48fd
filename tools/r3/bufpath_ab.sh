#!/bin/bash
# A/B: row fetches, compared pixels and mask stores through raw buffer resources (new) against 64-bit global addresses (prev)
cd "$(dirname "$0")/../.."
N=prlib_amd/libprlib_hip_testhooks.so
O=tools/probe/libprlib_hip_prev.so
for r in 1 2; do
bash tools/r3/ab.sh "--steps 20 --warmup 3" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method wolfjolion --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 64 --size 4096 --method sauvola --window 51 --k 0.34 --morph 0 --steps 10 --warmup 2" $O $N | tail -2
done
