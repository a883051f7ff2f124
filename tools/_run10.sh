cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fp32_split.py -x -q -k "ring" > gpurun_out/t_split_ring.txt 2>&1
tail -n 3 gpurun_out/t_split_ring.txt
MGN_FP32_SPLIT=4 timeout 300 python tools/ab.py default --rounds 2 > gpurun_out/ab_ring.txt 2>&1
tail -n 1 gpurun_out/ab_ring.txt
MGN_RING_GROUPS=1 MGN_FP32_SPLIT=4 timeout 300 python tools/ab.py default --rounds 1 > gpurun_out/ab_ring1.txt 2>&1
tail -n 1 gpurun_out/ab_ring1.txt
MGN_FP32_SPLIT=4 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/stamps.so python tools/diag_stamps_split.py > gpurun_out/stamps_ring.txt 2>&1
head -10 gpurun_out/stamps_ring.txt | cut -c1-330
