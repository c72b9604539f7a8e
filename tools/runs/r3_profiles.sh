#!/bin/bash
# round 3: the profiles that go into profiles/r03_* (kernel-trace + PMC of bench.py in the mode `value` is measured in; PMC traffic;
# every config of tools/bench_configs.py with its oracle check; per-kernel trace + traffic of the frame benchmark)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3p; mkdir -p $O
bash tools/profile.sh r03 > $O/profile_r03.log 2>&1
bash tools/pmc_traffic.sh r03 > $O/pmc_traffic_r03.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style_20_steps.json 2> $O/bench20.err
timeout 1500 python tools/bench_configs.py --iters 30 > $O/bench_configs.jsonl 2> $O/bench_configs.err
bash tools/kt.sh frame_r03 -- python tools/bench_configs.py --iters 20 --only frame_py > $O/kt_frame.txt 2>&1
bash tools/pmc_config.sh r03_frame frame_py > $O/pmc_frame.log 2>&1
bash tools/pmc_config.sh r03_4 4 > $O/pmc_4.log 2>&1
bash tools/pmc_config.sh r03_2 2_fused > $O/pmc_2.log 2>&1
tail -3 $O/bench_default.json | cut -c 1-300
