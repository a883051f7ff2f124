cd $GRAFT_REPO_ROOT
MGN_RING_GROUPS=1 MGN_FP32_SPLIT=4 timeout 300 python tools/ab.py default ring_nostag --rounds 2 > gpurun_out/ab_ring_stag.txt 2>&1
tail -n 2 gpurun_out/ab_ring_stag.txt
MGN_RING_EPI=1 MGN_RING_GROUPS=1 MGN_FP32_SPLIT=4 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/stamps_epi.so python tools/diag_stamps_split.py > gpurun_out/stamps_ring_epi.txt 2>&1
head -6 gpurun_out/stamps_ring_epi.txt | cut -c1-330
