#!/bin/bash
# round 3, GPU call 3: drawcull TASK emission with one lane per output command
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/kt.sh frame3 -- python tools/bench_configs.py --iters 10 --only frame_py > $O/kt_frame.txt 2>&1
grep -E "draw_|stats" $O/kt_frame.txt
timeout 600 python tools/bench_configs.py --iters 30 --only frame_py,3b,3b_fused,2 > $O/configs.jsonl 2> $O/configs.err
python3 - <<PY
import json
for l in open("$O/configs.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["config"][:60], {k: round(v, 1) for k, v in d.items() if k.endswith("_us")})
PY
