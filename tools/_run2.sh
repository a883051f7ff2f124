cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spec_variants.py -x -q 2>&1 | tail -30
python -m pytest tests/test_gpu_training_step.py tests/test_gpu_training_general.py tests/test_gpu_golden_and_partition.py -x -q 2>&1 | tail -5
