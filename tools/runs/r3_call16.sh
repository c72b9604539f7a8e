#!/bin/bash
# round 3, call 16: what bounds cluster_hiz_kernel and the direct-form cull kernels at frame scale — VALU issue or the texel gathers?
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/bound_frame; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --list-avail > $out/avail_full.txt 2>&1
grep -o -E "\b(SQ|TA|TCP|TCC|TD|GRBM|SPI)_[A-Za-z0-9_]+" $out/avail_full.txt | sort -u > $out/avail.txt
grep -o -E "\b(VALUBusy|SALUBusy|MemUnitBusy|MemUnitStalled|WriteUnitStalled|L2CacheHit|LDSBankConflict|VALUUtilization|FetchSize|WriteSize|MeanOccupancy[A-Za-z]*|[A-Z][A-Za-z]+Busy)\b" $out/avail_full.txt | sort -u > $out/avail_derived.txt
wc -l $out/avail.txt $out/avail_derived.txt
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_INSTS_SALU" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
           "VALUBusy MemUnitBusy MemUnitStalled L2CacheHit" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum"; do
  i=$((i+1))
  ( cd $R && timeout 400 rocprofv3 --kernel-trace --pmc $pmc -f csv -d $out/p$i -- python tools/bench_configs.py --iters 8 --only frame_py > $out/p$i.log 2>&1 ) || echo "pass $i failed: $pmc; $(grep -i -m3 "error\|invalid\|not found" $out/p$i.log)"
done
python3 - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$out/p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            if any(x in k for x in ("cluster_hiz", "cluster_mask", "draw_decide")):
                print(os.path.basename(d), "%-60s" % k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
