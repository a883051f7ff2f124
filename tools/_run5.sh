cd $GRAFT_REPO_ROOT
tools/_pl_probe
timeout 900 python -m pytest tests/test_gpu_fp32_split.py -x -q -k "ws or host" > gpurun_out/t_split_ws.txt 2>&1
tail -n 3 gpurun_out/t_split_ws.txt
MGN_FP32_SPLIT=3 timeout 300 python tools/ab.py default --rounds 2 > gpurun_out/ab_ws.txt 2>&1
tail -n 1 gpurun_out/ab_ws.txt
MGN_FP32_SPLIT=3 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/stamps.so python tools/diag_stamps_split.py > gpurun_out/stamps_ws.txt 2>&1
head -4 gpurun_out/stamps_ws.txt | tail -2
