#!/bin/bash
# A/B: k_fused held to 96 VGPRs (amdgpu_waves_per_eu(5): only NICK's instantiation is above, 98 -> 96 with one spilled register)
cd "$(dirname "$0")/../.."
N=tools/probe/libprlib_hip_w5.so
O=prlib_amd/libprlib_hip_testhooks.so
for r in 1 2; do
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 31 --k -0.01 --morph 0 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 101 --k -0.01 --morph 0 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--steps 20 --warmup 3" $O $N | tail -2
done
