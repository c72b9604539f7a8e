#!/bin/bash
# tools/build_rev.sh <git-rev> [name] — the product library of an earlier revision as an A/B baseline: variants/<name>.so (default name = the short
# hash; variants/ is git-ignored, travels with gpurun; use through NV_LIBRARY_PATH or tools/ab.sh).  Every variants/*.so DESIGN.md / EXPERIMENTS.md /
# profiles/ name is reproducible with one such line, e.g.
#   tools/build_rev.sh 58a58d0 base_r5     # the tree round 5's second session started from
#   tools/build_rev.sh 95dfdfd r4          # round 4's final tree
set -e
rev=$1; name=${2:-$(git rev-parse --short "$1")}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/nv_rev_XXXXXX)
git -C "$root" archive "$rev" | tar -x -C "$tmp"
make -s -C "$tmp/niagara_amd/csrc" -j8 ../libniagara_vis.so
mkdir -p "$root/variants"
cp "$tmp/niagara_amd/libniagara_vis.so" "$root/variants/$name.so"
rm -rf "$tmp"
echo "variants/$name.so  ($(git -C "$root" rev-parse --short "$rev"))"
