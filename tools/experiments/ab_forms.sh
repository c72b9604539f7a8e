#!/bin/bash
# usage (through gpurun): bash tools/experiments/ab_forms.sh [-r rounds] [-c configs] form ... — round 6: the cull launch's forms (NV_OPT_CULL_FORM pinned through
# NV_BENCH_CULL_FORM: 0 = the host's choice, 3 = direct with the packed walk where it applies, 4 = direct, one command per wave iteration) on the product
# library, interleaved on one box: the contract chain at BASELINE scale, then tools/bench_configs.py lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rounds=2; cfgs="3a_dense,frame_py"; chain=1
while getopts "r:c:n" o; do case $o in r) rounds=$OPTARG;; c) cfgs=$OPTARG;; n) chain=0;; esac; done
shift $((OPTIND - 1))
for round in $(seq $rounds); do
for form in "$@"; do
  echo "== NV_OPT_CULL_FORM $form (round $round)"
  if [ $chain = 1 ]; then
  NV_BENCH_CULL_FORM=$form timeout 600 python - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import bench
r = bench.contract_chain(0)
print("contract_chain", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_us") or k in ("us_per_phase", "parity")}, "frac", round(r["roofline"]["frac"], 3))
PY
  fi
  if [ -n "$cfgs" ]; then
  NV_BENCH_CULL_FORM=$form timeout 900 python tools/bench_configs.py --iters 60 --only $cfgs 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','parity','kernel_variants')})"
  fi
done
done
