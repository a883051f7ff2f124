cd $GRAFT_REPO_ROOT
MGN_RING_HP=1 python -m pytest tests/test_gpu_fp32_split.py -x -q -k "edge_kernel_meets or ragged" 2>&1 | tail -4
python tools/ab.py default default+MGN_RING_HP=1 --rounds 2 2>&1 | tail -3
