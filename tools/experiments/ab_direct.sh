#!/bin/bash
# usage (through gpurun): bash tools/experiments/ab_direct.sh [-r rounds] [-c configs] lib.so ... — round 6: A/B of library builds on the configs whose
# cluster launch takes the DIRECT form (contract chain at BASELINE scale = bench.py's `contract_chain`, 3A dense, the frame), interleaved on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rounds=2; cfgs="3a_dense,frame_py"; chain=1
while getopts "r:c:n" o; do case $o in r) rounds=$OPTARG;; c) cfgs=$OPTARG;; n) chain=0;; esac; done
shift $((OPTIND - 1))
for round in $(seq $rounds); do
for so in "$@"; do
  echo "== $so (round $round)"
  if [ $chain = 1 ]; then
  NV_LIBRARY_PATH=$R/$so timeout 600 python - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import bench
r = bench.contract_chain(0)
print("contract_chain", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_us") or k in ("us_per_phase", "parity")}, "frac", round(r["roofline"]["frac"], 3))
PY
  fi
  if [ -n "$cfgs" ]; then
  NV_LIBRARY_PATH=$R/$so timeout 900 python tools/bench_configs.py --iters 60 --only $cfgs 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','parity')})"
  fi
done
done
