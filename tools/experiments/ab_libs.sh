#!/bin/bash
# usage (through gpurun): bash tools/experiments/ab_libs.sh [-r rounds] [-c configs] [-t] lib ... — A/B of whole libraries (tools/build_variant.sh; "product" = the
# library as built) interleaved on one box: tools/bench_configs.py lines per library; -t first runs the drawcull-facing GPU tests against each library.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rounds=2; cfgs="2,2L,frame"; tests=0
while getopts "r:c:t" o; do case $o in r) rounds=$OPTARG;; c) cfgs=$OPTARG;; t) tests=1;; esac; done
shift $((OPTIND - 1))
lib_env() { if [ "$1" = product ]; then unset NV_LIBRARY_PATH; else export NV_LIBRARY_PATH=$R/variants/$1.so; fi; }
if [ $tests = 1 ]; then
for lib in "$@"; do
  [ $lib = product ] && continue
  lib_env $lib; echo "== tests against $lib"
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_frame_driver.py -m gpu -x -q -k "draw or frame or config or fuzz" 2>&1 | tail -4
done
fi
for round in $(seq $rounds); do
for lib in "$@"; do
  lib_env $lib; echo "== $lib (round $round)"
  timeout 900 python tools/bench_configs.py --iters 60 --only $cfgs 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','parity')})"
done
done
