cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fp32_split.py -x -q > gpurun_out/t_split_all.txt 2>&1
tail -n 6 gpurun_out/t_split_all.txt
for m in 2 3; do MGN_FP32_SPLIT=$m timeout 300 python tools/ab.py default --rounds 2 > gpurun_out/ab_node_$m.txt 2>&1; tail -n 1 gpurun_out/ab_node_$m.txt; done
