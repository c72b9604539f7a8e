#!/bin/bash
# round 3, call 30: the whole GPU suite on the final tree, config T's line for profiles/
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_distributed_gpu.py > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 400 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_dist.log 2>&1; tail -3 $O/pytest_dist.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench20.json; cut -c1-330 $O/bench20.json
timeout 300 python tools/bench_configs.py --iters 30 --only task 2>/dev/null | grep "^{" > $O/task.jsonl; cut -c1-300 $O/task.jsonl
