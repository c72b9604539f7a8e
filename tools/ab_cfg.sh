#!/bin/bash
# usage: bash tools/ab_cfg.sh "<modes>" "<configs>" — tools/bench_configs.py per NV_DEBUG_MODE
for m in $1; do
  NV_DEBUG_MODE=$m python tools/bench_configs.py --only $2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('mode', $m, d['config'][:12], {k: round(v,2) for k,v in d.items() if k in ('late_cull_us','cull_us','cluster_cull_us','step_us')})"
done
