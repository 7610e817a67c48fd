#!/bin/bash
# A/B: the slide of the window (and the next row's fetches) moved up behind the issue of the bpermutes (se1), pinned there with a sched_barrier (se2)
cd "$(dirname "$0")/../.."
L="prlib_amd/libprlib_hip_testhooks.so tools/probe/libprlib_hip_se1.so tools/probe/libprlib_hip_se2.so"
bash tools/r3/ab.sh "--steps 20 --warmup 3" $L
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2" $L
