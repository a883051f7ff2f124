cd $GRAFT_REPO_ROOT
MGN_FP32_SPLIT=2 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/stamps.so python tools/diag_stamps_split.py > gpurun_out/stamps_split2.txt 2>&1
cat gpurun_out/stamps_split2.txt | cut -c1-400
