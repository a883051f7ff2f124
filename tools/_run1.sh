cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_training_step.py tests/test_gpu_training_general.py tests/test_gpu_hidden_layers.py -x -q 2>&1 | tail -3
echo "== auto"; timeout 600 python tools/step_1m.py 2>&1 | tail -4
echo "== recompute all"; MGN_TRAIN_RECOMPUTE=1 timeout 600 python tools/step_1m.py 2>&1 | tail -4
echo "== keep 8"; MGN_TRAIN_KEEP_STEPS=8 timeout 600 python tools/step_1m.py 2>&1 | tail -4
