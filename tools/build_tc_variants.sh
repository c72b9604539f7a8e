#!/bin/bash
# tools/build_tc_variants.sh — A/B builds of the product library that differ only in trianglecull.hip's batch capacities / grid
# (variants/tc_<VCAP>_<TCAP>_<blocks per CU>.so, git-ignored).  usage: bash tools/build_tc_variants.sh "VCAP TCAP blocks-per-CU [chunk [extra macro]]" ...
set -e
cd "$(dirname "$0")/../niagara_amd/csrc"
make -s ../libniagara_vis.so
mkdir -p ../../variants build/tcv
for v in "$@"; do
  set -- $v
  c=${4:-64}; x=${5:-}
  n=tc_$1_$2_$3_$c$x
  o=build/tcv/$n.o
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function -DTC_VCAP=$1 -DTC_TCAP=$2 -DTC_BLOCKS_PER_CU=$3 -DTC_CHUNK=$c ${x:+-D$x} -c trianglecull.hip -o $o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$n.so build/clustercull.o build/drawcull.o build/submit.o build/depthreduce.o $o build/bounds.o build/context.o build/host.o
  echo variants/$n.so
done
