cd $GRAFT_REPO_ROOT
python tools/ab.py default wh1 wh2 wh4 wh8 wh16 wh32 wh12 wh3 wh63 --rounds 2 2>&1 | tail -12
