#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config'][:50], d['parity'], {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})"; }
timeout 600 python tools/bench_configs.py --iters 30 --only 3a_dense,3a,frame_py 2>/dev/null | show
bash tools/kt.sh sc -- python tools/bench_configs.py --iters 20 --only 3a_dense,frame_py 2>&1 | grep -E "scatter|stats"
