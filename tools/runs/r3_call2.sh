#!/bin/bash
# round 3, GPU call 2: where the frame's time goes (kernel-trace + instruction counters), occlusion-stage knobs, pyramid modes
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
EXP=$PWD/niagara_amd/libniagara_vis_exp.so
bash tools/kt.sh frame -- python tools/bench_configs.py --iters 10 --only frame_py > $O/kt_frame.txt 2>&1
tail -30 $O/kt_frame.txt
for kv in "NV_HIZ_SHARERS=2" "NV_HIZ_SHARERS=8" "NV_HIZ_LIST_STRIDE=64" "NV_HIZ_MIN_PER=32" "NV_HIZ_SHARERS=8 NV_HIZ_MIN_PER=16"; do
  echo "== $kv" >> $O/hiz_knobs.txt
  env NV_LIBRARY_PATH=$EXP $kv timeout 300 python tools/bench_configs.py --iters 10 --only frame_py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: round(v, 1) for k, v in d.items() if k.endswith('_us')})" >> $O/hiz_knobs.txt
done
cat $O/hiz_knobs.txt
for m in 0 1; do bash tools/kt.sh pyr$m NV_LIBRARY_PATH=$EXP NV_PYRAMID_MODE=$m -- python tools/bench_configs.py --iters 30 --only 4 > $O/kt_pyr$m.txt 2>&1; grep -E "reduce|stats" $O/kt_pyr$m.txt; done
bash tools/pmc.sh frame1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" -- python tools/bench_configs.py --iters 5 --only frame_py > $O/pmc_frame1.txt 2>&1
grep -E "pmc|cluster|draw_" $O/pmc_frame1.txt | cut -c 1-400
bash tools/pmc.sh frame2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" -- python tools/bench_configs.py --iters 5 --only frame_py > $O/pmc_frame2.txt 2>&1
grep -E "pmc|cluster|draw_" $O/pmc_frame2.txt | cut -c 1-400
