#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3o; mkdir -p $O
show() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config'][:70], d['parity'], {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})"; }
bash tools/kt.sh f15 -- python tools/bench_configs.py --iters 20 --only frame_py,3b_fused 2>&1 | grep -E "draw_|stats"
timeout 600 python tools/bench_configs.py --iters 30 --only 3b_fused,frame_py 2>/dev/null | show
timeout 600 python -m pytest tests -m gpu -x -q --deselect tests/test_distributed_gpu.py > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 400 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_dist.log 2>&1; tail -15 $O/pytest_dist.log
