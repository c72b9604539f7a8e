#!/bin/bash
# usage (through gpurun): bash tools/experiments/ab_task_emit.sh [-r rounds] emit ... — round 6: nv_drawcull(task = 1)'s emission forms (NV_OPT_TASK_EMIT pinned through
# NV_BENCH_TASK_EMIT: 0 = by statistic, 1 = per draw, 2 = the list form fed by the decide launch's records) on the contract chain and config 3B / the frame
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rounds=2
while getopts "r:" o; do case $o in r) rounds=$OPTARG;; esac; done
shift $((OPTIND - 1))
for round in $(seq $rounds); do
for emit in "$@"; do
  echo "== NV_OPT_TASK_EMIT $emit (round $round)"
  NV_BENCH_TASK_EMIT=$emit timeout 600 python - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import bench
r = bench.contract_chain(0)
print("contract_chain", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_us") or k in ("us_per_phase", "parity")}, "frac", round(r["roofline"]["frac"], 3))
PY
  NV_BENCH_TASK_EMIT=$emit timeout 900 python tools/bench_configs.py --iters 60 --only 3b,3b_fused,frame 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','parity')})"
done
done
