#!/bin/bash
# A/B: s1 = iterative-minreg, s2 = machine scheduler off, s3 = schedule-metric-bias 100, s4 = pre- and post-RA schedulers off
cd "$(dirname "$0")/../.."
L="prlib_amd/libprlib_hip_testhooks.so tools/probe/libprlib_hip_s1.so tools/probe/libprlib_hip_s2.so tools/probe/libprlib_hip_s3.so tools/probe/libprlib_hip_s4.so"
bash tools/r3/ab.sh "--steps 20 --warmup 3" $L
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2" $L
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2" $L
