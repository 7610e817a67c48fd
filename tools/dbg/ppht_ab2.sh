#!/bin/bash
# tools/dbg/ppht_ab2.sh <tag> : default hooks lib, XCD placement on / off
tag=$1
for x in 1 0; do
  echo "==== PRL_HIP_PPHT_GROUP_XCD=$x"
  PRL_HIP_PPHT_GROUP_XCD=$x timeout 300 python tools/dbg/ppht_real.py 64 > gpurun_out/ab_${tag}_xcd${x}_real.txt 2>&1
  grep "visiting\|^real\|^synth\|one page" gpurun_out/ab_${tag}_xcd${x}_real.txt | cut -c1-110
  PRL_HIP_PPHT_GROUP_XCD=$x timeout 300 python tools/dbg/ppht_group_prof.py 1 256 > gpurun_out/ab_${tag}_xcd${x}_text.txt 2>&1
  grep "deskew_s" gpurun_out/ab_${tag}_xcd${x}_text.txt
done
