#!/bin/bash
# round 4, call 5: DPP wave scans instead of ds_bpermute shuffles (scatter launches, lane kernels), lane late forms removed: full GPU suite,
# then the headline and the configs against round 3's library on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
bash tools/ab.sh -r 2 -s 200 -c 2_fused,3b_fused,frame_py,3a_dense variants/r3.so niagara_amd/libniagara_vis.so 2>&1 | tee $O/ab.txt
