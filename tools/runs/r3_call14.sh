#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config'][:70], d['parity'], {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/kt.sh f14 -- python tools/bench_configs.py --iters 20 --only frame_py,3b_fused 2>&1 | grep -E "draw_|stats"
timeout 600 python tools/bench_configs.py --iters 30 --only 3b_fused,frame_py 2>/dev/null | show
