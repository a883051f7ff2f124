cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_spec_variants.py tests/test_gpu_parity.py tests/test_gpu_golden_and_partition.py -x -q > gpurun_out/t_spec.txt 2>&1
tail -n 12 gpurun_out/t_spec.txt
