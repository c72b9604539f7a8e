#!/bin/bash
# tools/build_variant.sh — an A/B build of the product library with ONE source file compiled with extra macros:
#   bash tools/build_variant.sh <name> <file without .hip> [-DMACRO=V ...]   ->  variants/<name>.so (git-ignored; NV_LIBRARY_PATH)
# (clustercull.hip variants: tools/build_cc_variants.sh also knows the experiments flavour; the ISA hazard scan belongs to `make`, not to these)
set -e
cd "$(dirname "$0")/../niagara_amd/csrc"
make -s ../libniagara_vis.so
mkdir -p ../../variants build/var
n=$1; f=$2; shift 2
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $f.hip -o build/var/$n.o
objs=""
for s in clustercull drawcull submit depthreduce trianglecull bounds context; do
  if [ $s = $f ]; then objs="$objs build/var/$n.o"; else objs="$objs build/$s.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$n.so $objs build/host.o
echo variants/$n.so
