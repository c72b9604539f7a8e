#!/bin/bash
# tools/build_cc_variants.sh — A/B builds of the product library that differ only in clustercull.hip's macros
# (variants/cc_<name>.so, git-ignored).  usage: bash tools/build_cc_variants.sh "name -DMACRO=V ..." ...
# A name that starts with e_ is built from the experiments flavour (-DNV_EXPERIMENTS: NV_DEBUG_MODE and the NV_* environment knobs work).
# (the ISA hazard scan is part of `make`, not of these builds: use them for timing A/Bs of changes outside the inline-asm rings only)
set -e
cd "$(dirname "$0")/../niagara_amd/csrc"
make -s ../libniagara_vis.so ../libniagara_vis_exp.so
mkdir -p ../../variants build/ccv
for v in "$@"; do
  set -- $v
  n=cc_$1; shift
  o=build/ccv/$n.o
  b=build; x=
  case $n in cc_e_*) b=build/exp; x=-DNV_EXPERIMENTS;; esac
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function $x "$@" -c clustercull.hip -o $o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$n.so $o $b/drawcull.o $b/submit.o $b/depthreduce.o $b/trianglecull.o $b/bounds.o $b/context.o build/host.o
  echo variants/$n.so
done
