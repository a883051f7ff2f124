cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spec_variants.py tests/test_gpu_training_step.py -x -q 2>&1 | tail -3
python tools/lnall_timing.py 2>&1 | tail -2
