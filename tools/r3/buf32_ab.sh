cd /root/repo
N=tools/probe/libprlib_hip_buf32.so
O=prlib_amd/libprlib_hip_testhooks.so
for r in 1 2; do
bash tools/r3/ab.sh "--pages 256 --size 2480 --height 3508 --method niblack --window 101 --k 0.01 --morph 2 --steps 10 --warmup 2" $O $N | tail -2
bash tools/r3/ab.sh "--pages 64 --size 4096 --method sauvola --window 51 --k 0.34 --morph 0 --steps 10 --warmup 2" $O $N | tail -2
done
