for b in 3 4 5 6 7 8; do
NV_CC_BLOCKS_PER_CU=$b python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_avg_us'],2), round(d['roofline']['scatter_kernel_avg_us'],2))"
done
