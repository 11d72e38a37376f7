# round 4: where k_row16hk should take over from the per-board path in a SEARCH (tree step + forward, 400 sims): move decisions/s by number of games
for g in 16 24 32 48 64 96 128 192 256 384 512; do
  for cfg in "0,-1 7776" "1,47 7776" "1,47 2592" "1,47 1296"; do
    set -- $cfg
    echo -n "games $g AO_ROWK=$1 AO_PERBOARD_CELLS=$2: "; AO_ROWK=$1 AO_PERBOARD_CELLS=$2 python tools/time_single_game.py --moves 4 --games $g 2>&1 | grep "us/sim"
  done
done
