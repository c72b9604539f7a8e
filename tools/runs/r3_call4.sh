#!/bin/bash
# round 3, GPU call 4: the scatter riding in the cull launch
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "riding" > $O/pytest_riding.log 2>&1; echo "pytest rc $?" >> $O/pytest_riding.log
tail -15 $O/pytest_riding.log
for r in 0 1 0 1; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --overlap-streams 0 --riding-scatter $r > $O/bench_r$r.json 2> $O/bench_r$r.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench_r$r.json').read().strip().splitlines()[-1]); r=d['roofline']
print('riding $r: ms_per_step %.5f value %.1f G/s kernel %.2f us scatter %.2f pass_frac %.3f visible %d' % (d['ms_per_step'], d['value']/1e9, r['kernel_avg_us'], r['scatter_kernel_avg_us'], r['pass_frac'], d['config']['visible_total']))"
done
bash tools/kt.sh ride1 -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 --riding-scatter 1 > $O/kt_ride1.txt 2>&1; grep -E "cluster|stats" $O/kt_ride1.txt
bash tools/kt.sh ride0 -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-streams 0 --riding-scatter 0 > $O/kt_ride0.txt 2>&1; grep -E "cluster|stats" $O/kt_ride0.txt
