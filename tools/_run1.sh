cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/_overlap_probe > gpurun_out/overlap_probe.txt 2>&1
MGN_FP32_SPLIT=1 python tools/ab.py default wi_lds wi_nosplit wi_both --rounds 2 > gpurun_out/ab_whatif.txt 2>&1
MGN_FP32_SPLIT=1 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/stamps.so python tools/diag_stamps_split.py > gpurun_out/stamps_split.txt 2>&1
tail -5 gpurun_out/ab_whatif.txt
