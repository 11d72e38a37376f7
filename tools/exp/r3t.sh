cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo -n "1 game  "; python tools/time_single_game.py --moves 12 2>&1 | grep "us/sim"
done
for g in 8 24 48; do echo -n "games $g  "; python tools/time_single_game.py --moves 6 --games $g 2>&1 | grep "us/sim"; done
AO_LIB_TAG=prof AO_PROF_PRINT=1 AO_FUSED_STEP=0 python tools/time_single_game.py --moves 1 2>&1 | grep "AO_PROF k_conv" | tail -2
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -3
