#!/bin/bash
cfg=$1; shift
for so in "$@"; do
  NV_LIBRARY_PATH=$PWD/$so python tools/bench_configs.py --only $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('$so', d['config'][:12], {k: round(v,2) for k,v in d.items() if k in ('late_cull_us','cull_us','cluster_cull_us','step_us','kernel_us')})"
done
