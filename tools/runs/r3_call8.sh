#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "taskcull or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/kt.sh task -- python tools/bench_configs.py --iters 40 --only task > $O/kt_task.txt 2>&1; grep -E "taskcull|stats" $O/kt_task.txt; grep -o '"call_us": [0-9.]*, "algorithmic_bytes": [0-9]*, "achieved_GBs": [0-9.]*, "frac": [0-9.]*' gpurun_out/kt_task/run.log
timeout 300 python tools/bench_configs.py --iters 40 --only task | cut -c 1-400
