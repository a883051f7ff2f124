cd $GRAFT_REPO_ROOT
python tools/ab.py default default+MGN_RING_WAVES=4 --rounds 2 2>&1 | tail -3
