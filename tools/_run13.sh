cd $GRAFT_REPO_ROOT
MGN_RING_GROUPS=1 MGN_FP32_SPLIT=4 timeout 600 python tools/ab.py default ring_norf ring_rot1 --rounds 3 > gpurun_out/ab_ring_cmp.txt 2>&1
tail -n 3 gpurun_out/ab_ring_cmp.txt
MGN_FP32_SPLIT=2 timeout 300 python tools/ab.py default --rounds 2 > gpurun_out/ab_split2.txt 2>&1
tail -n 1 gpurun_out/ab_split2.txt
MGN_RING_GROUPS=1 MGN_FP32_SPLIT=4 MGN_LIB_PATH=$GRAFT_REPO_ROOT/meshgraphnets.jl_amd/lib/variants/ring_norf.so timeout 600 python -m pytest tests/test_gpu_fp32_split.py -x -q -k ring 2>&1 | tail -2
