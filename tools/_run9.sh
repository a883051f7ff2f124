cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/t_gpu_all.txt 2>&1
tail -n 30 gpurun_out/t_gpu_all.txt
