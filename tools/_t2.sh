cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spec_variants.py -x -q 2>&1 | tail -5
