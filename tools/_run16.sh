cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fp32_split.py -x -q > gpurun_out/t_split.txt 2>&1
tail -n 4 gpurun_out/t_split.txt
timeout 300 python tools/ab.py default --rounds 2 > gpurun_out/ab_noderings.txt 2>&1
tail -n 1 gpurun_out/ab_noderings.txt
MGN_FP32_SPLIT=2 timeout 300 python tools/ab.py default --rounds 1 > gpurun_out/ab_mode2.txt 2>&1
tail -n 1 gpurun_out/ab_mode2.txt
