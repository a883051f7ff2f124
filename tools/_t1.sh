cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fp32_split.py -x -q -k "any_scale" 2>&1 | tail -15
