#!/bin/bash
# Experiment builds of the library with parts of k_fused's float32 row loop cut out (speed-of-light probes):
# the probe variants live in k_fused_probes.patch, NOT in the product source.  This script applies the patch to a copy of
# prlib_amd/csrc and builds tools/probe_build/libprlib_probe<N>.so for every N given (default 1..6).
#   tools/probe/build_probes.sh [N ...]      then: tools/probe_fused.sh on the GPU box
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
WORK=$ROOT/build_probe_src
rm -rf "$WORK" && mkdir -p "$WORK/prlib_amd" "$ROOT/tools/probe_build"
cp -r prlib_amd/csrc "$WORK/prlib_amd/csrc" && cp -r include "$WORK/include"
rm -rf "$WORK"/prlib_amd/csrc/build*
patch -s -p0 -d "$WORK" prlib_amd/csrc/binarize_fused.hip < tools/probe/k_fused_probes.patch
for n in ${@:-1 2 3 4 5 6}; do
  make -C "$WORK/prlib_amd/csrc" -s -j8 OBJDIR=build_probe$n OUT="$ROOT/tools/probe_build/libprlib_probe$n.so" EXTRA="-DPRL_PROBE=$n -DPRL_TEST_HOOKS"
done
ls -la "$ROOT"/tools/probe_build/
