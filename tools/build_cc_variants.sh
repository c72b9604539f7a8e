#!/bin/bash
# tools/build_cc_variants.sh — A/B builds of the product library that differ only in clustercull.hip's macros
# (variants/cc_<name>.so, git-ignored).  usage: bash tools/build_cc_variants.sh "name -DMACRO=V ..." ...
# (the ISA hazard scan is part of `make`, not of these builds: use them for timing A/Bs of changes outside the inline-asm rings only)
set -e
cd "$(dirname "$0")/../niagara_amd/csrc"
make -s ../libniagara_vis.so
mkdir -p ../../variants build/ccv
for v in "$@"; do
  set -- $v
  n=cc_$1; shift
  o=build/ccv/$n.o
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c clustercull.hip -o $o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$n.so $o build/drawcull.o build/submit.o build/depthreduce.o build/trianglecull.o build/bounds.o build/context.o build/host.o
  echo variants/$n.so
done
