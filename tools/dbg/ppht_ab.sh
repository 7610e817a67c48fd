#!/bin/bash
# A/B of group-kernel builds: tools/dbg/ppht_ab.sh <tag> [libs...]  (each lib: check vs mw on small cases is done once with the default)
tag=$1; shift
for lib in "$@"; do
  echo "==== $lib"
  PRL_LIB=$lib timeout 300 python tools/dbg/ppht_real.py 64 > gpurun_out/ab_${tag}_$(basename $lib .so)_real.txt 2>&1
  grep "visiting\|^real\|^synth\|one page" gpurun_out/ab_${tag}_$(basename $lib .so)_real.txt | cut -c1-120
  PRL_LIB=$lib timeout 300 python tools/dbg/ppht_group_prof.py 1 256 > gpurun_out/ab_${tag}_$(basename $lib .so)_text.txt 2>&1
  grep "deskew_s" gpurun_out/ab_${tag}_$(basename $lib .so)_text.txt
done
