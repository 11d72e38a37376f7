cd $GRAFT_REPO_ROOT
for g in 16 24 36 48; do for m in 3 6; do echo -n "15x15 games $g mode $m  "; python tools/time_single_game.py --moves 3 --board 15 --games $g --mode $m --sims 200 2>&1 | grep "us/sim"; done; done
for g in 80 100 112; do for m in 3 6; do echo -n "9x9 games $g mode $m  "; python tools/time_single_game.py --moves 4 --games $g --mode $m 2>&1 | grep "us/sim"; done; done
