#!/bin/bash
# usage (through gpurun): bash tools/experiments/ab_chain_libs.sh [-r rounds] lib ... — bench.py's contract chain (BASELINE configs[2] with LOD select) per library
# ("product" or a variants/ name), interleaved on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rounds=3
while getopts "r:" o; do case $o in r) rounds=$OPTARG;; esac; done
shift $((OPTIND - 1))
for round in $(seq $rounds); do
for lib in "$@"; do
  if [ "$lib" = product ]; then unset NV_LIBRARY_PATH; else export NV_LIBRARY_PATH=$R/variants/$lib.so; fi
  timeout 600 python - <<PY 2>/dev/null
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import bench
r = bench.contract_chain(0)
print("$lib", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_us") or k in ("us_per_phase", "parity")})
PY
done
done
