#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config'][:50], d['parity'], {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})"; }
timeout 600 python tools/bench_configs.py --iters 30 --only frame_py,3b_fused 2>/dev/null | show
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
