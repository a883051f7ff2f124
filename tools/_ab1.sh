cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fp32_split.py tests/test_gpu_renumber.py -x -q 2>&1 | tail -4
python tools/ab.py default ahead2 --rounds 3 2>&1 | tail -6
