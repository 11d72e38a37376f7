cd $GRAFT_REPO_ROOT
python tools/time_self_play.py 4096 400 4 1 -3 2>&1 | grep -v amdgpu
python tools/time_self_play.py 4096 400 4 1 0 2>&1 | grep -v amdgpu | head -3
