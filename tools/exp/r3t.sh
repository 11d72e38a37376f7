cd $GRAFT_REPO_ROOT
AO_LIB_TAG=prof AO_PROF_TREE=1 python tools/time_single_game.py --moves 2 2>&1 | grep "AO_PROF k_step" | tail -2
for i in 1 2; do
for f in 1 0; do
echo -n "AO_FUSED_STEP=$f  "; AO_FUSED_STEP=$f python tools/time_single_game.py --moves 12 2>&1 | grep "us/sim"
done; done
for g in 8 24 48; do for f in 1 0; do
echo -n "games $g AO_FUSED_STEP=$f  "; AO_FUSED_STEP=$f python tools/time_single_game.py --moves 6 --games $g 2>&1 | grep "us/sim"
done; done
python -m pytest tests/test_gpu_net.py tests/test_gpu_dropin.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3
