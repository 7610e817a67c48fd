cd /root/repo
bash tools/r3/ab.sh "--steps 20 --warmup 3" prlib_amd/libprlib_hip_testhooks.so tools/probe/libprlib_hip_bufpv.so
bash tools/r3/ab.sh "--steps 20 --warmup 3" prlib_amd/libprlib_hip_testhooks.so tools/probe/libprlib_hip_bufpv.so
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method nick --window 21 --k -0.01 --morph 0 --steps 10 --warmup 2" prlib_amd/libprlib_hip_testhooks.so tools/probe/libprlib_hip_bufpv.so
