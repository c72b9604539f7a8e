#!/bin/bash
# usage: build_variant.sh name "-DFLAG=..."   (working tree, product library only)
set -e
name=$1; flags=$2
tmp=$(mktemp -d /tmp/nv_var_XXXXXX)
cp -r /root/repo/niagara_amd /root/repo/include /root/repo/tools $tmp/ 2>/dev/null
rm -rf $tmp/niagara_amd/csrc/build $tmp/niagara_amd/*.so
make -s -C $tmp/niagara_amd/csrc -j4 ../libniagara_vis.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function $flags" 2>&1 | grep -v "^inline-asm\|^in-flight\|^counted" || true
mkdir -p /root/repo/variants
cp $tmp/niagara_amd/libniagara_vis.so /root/repo/variants/$name.so
rm -rf $tmp
echo "variants/$name.so"
