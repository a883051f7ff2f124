cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/t_gpu_all.txt 2>&1
tail -n 8 gpurun_out/t_gpu_all.txt
python tools/ab.py default --rounds 2 > gpurun_out/ab_default.txt 2>&1; tail -n 1 gpurun_out/ab_default.txt
