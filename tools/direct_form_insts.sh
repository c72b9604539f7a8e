#!/bin/bash
# per-config instruction counters of the direct form's cull launch, one command per iteration (NV_OPT_CULL_FORM 4) against the packed walk (0): separate rocprofv3 --pmc passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r06}
out=$R/gpurun_out/insts_${TAG}; mkdir -p $out
for cfg in 3b_chain 3a_dense frame_py; do for form in 4 0; do
  bash tools/pmc.sh ${TAG}i_${cfg}_$form "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" NV_BENCH_CULL_FORM=$form -- python tools/bench_configs.py --iters 20 --only $cfg 2>&1 | grep "cluster_mask_kernel\|cluster_bits" > $out/${cfg}_form$form.txt
done
bash tools/pmc.sh ${TAG}c1_${cfg} "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" -- python tools/bench_configs.py --iters 20 --only $cfg 2>&1 | grep "cluster_mask_kernel\|cluster_bits\|cluster_hiz" > $out/${cfg}_cls1.txt
bash tools/pmc.sh ${TAG}c2_${cfg} "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" -- python tools/bench_configs.py --iters 20 --only $cfg 2>&1 | grep "cluster_mask_kernel\|cluster_bits\|cluster_hiz" > $out/${cfg}_cls2.txt
done
