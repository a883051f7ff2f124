cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden_and_partition.py tests/test_gpu_baseline_shapes.py tests/test_gpu_fp32_split.py tests/test_gpu_two_edge_sets.py tests/test_gpu_renumber.py tests/test_gpu_random_sweep.py -x -q 2>&1 | tail -5
python tools/fwd_breakdown.py 2>&1 | grep -v amdgpu | tail -4
python tools/rhs_breakdown.py 2>&1 | grep -v amdgpu | tail -6
