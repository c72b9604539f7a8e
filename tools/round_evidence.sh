#!/bin/bash
# round 6 evidence in one gpurun call: tools/round_numbers.sh ${TAG}, per-kernel traffic of the frame / config 2 / config 4, instruction counters of the
# direct form before / after the packed walk (item 2), the plain-loads build's lines (item 6d), the counters this rocprofv3 offers for an executed class mix (item 6c)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r06}
out=$R/gpurun_out/evidence_${TAG}; mkdir -p $out
bash tools/round_numbers.sh ${TAG} > $out/round_numbers.txt 2>&1
for cfg in frame_py 2_fused 4 3b_chain 3a_dense; do
  bash tools/pmc_config.sh ${TAG}_$cfg $cfg > $out/pmc_config_$cfg.txt 2>&1; cp gpurun_out/pmcc_${TAG}_$cfg/summary.json $out/pmc_traffic_$cfg.json 2>/dev/null
done
for form in 4 0; do
  bash tools/pmc.sh ${TAG}insts_form$form "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" NV_BENCH_CULL_FORM=$form -- python tools/bench_configs.py --iters 20 --only 3b_chain,frame_py,3a_dense > $out/pmc_insts_form$form.txt 2>&1
done
bash tools/pmc_valu.sh ${TAG} > $out/pmc_valu.txt 2>&1; cp gpurun_out/pmcv_${TAG}/valu_counters.json $out/valu_counters.json 2>/dev/null
( NV_LIBRARY_PATH=$R/niagara_amd/libniagara_vis_plain.so python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > $out/plain_bench.json
  NV_LIBRARY_PATH=$R/niagara_amd/libniagara_vis_plain.so python tools/bench_configs.py --iters 60 --only 3b_chain,frame_py,3a_dense 2>/dev/null | grep "^{" > $out/plain_configs.jsonl
  python tools/bench_configs.py --iters 60 --only 3b_chain,frame_py,3a_dense 2>/dev/null | grep "^{" > $out/asm_configs.jsonl ) 
(rocprofv3-avail list 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -i "SQ_INST\|SQ_ACTIVE\|SQ_VALU\|TRANS\|MFMA" | head -80 > $out/counters_avail.txt
ls $out
