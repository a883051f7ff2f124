cd $GRAFT_REPO_ROOT
for k in 0 4; do
echo "== stage $k"
MGN_WS_K=1 MGN_FP32_SPLIT=3 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/stamps_k$k.so python tools/diag_stamps_split.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/stamps_ws_k.txt
cat gpurun_out/stamps_ws_k.txt
