cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_training_step.py -x -q 2>&1 | tail -3
python tools/step_timing.py 2>&1 | grep "mgn_step median"
timeout 900 python tools/step_1m.py 2>&1 | tail -3
