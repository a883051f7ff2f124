cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fp32_split.py -x -q > gpurun_out/t_split.txt 2>&1
tail -n 5 gpurun_out/t_split.txt
MGN_FP32_SPLIT=1 python tools/ab.py default --rounds 1 > gpurun_out/ab_sp1.txt 2>&1
MGN_FP32_SPLIT=2 python tools/ab.py default sp2_d8 sp2_noil sp2_noslp --rounds 2 > gpurun_out/ab_sp2.txt 2>&1
tail -n 1 gpurun_out/ab_sp1.txt; tail -n 4 gpurun_out/ab_sp2.txt
