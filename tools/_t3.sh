cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_training_step.py tests/test_gpu_training_general.py -x -q 2>&1 | tail -3
python tools/step_timing.py 2>&1 | grep median
python tools/step_1m.py 2>&1 | tail -1
