cd $GRAFT_REPO_ROOT
python tools/fwd_breakdown.py 2>&1 | grep -v amdgpu | tail -5
