#!/bin/bash
# usage: bash tools/ab2.sh variants/libA.so variants/libB.so ... — bench.py (twice, interleaved) + configs 2, 3b, 4 per library build
for round in 1 2; do
for so in "$@"; do
  echo "== $so (bench, round $round)"
  NV_LIBRARY_PATH=$PWD/$so timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_avg_us'],2), round(d['roofline']['scatter_kernel_avg_us'],2), d['config']['visible_total'])"
done
done
for so in "$@"; do
  echo "== $so (configs)"
  NV_LIBRARY_PATH=$PWD/$so timeout 200 python tools/bench_configs.py --iters 100 --only 2,3b,4 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_us') or k in ('visible','late_visible')})"
done
