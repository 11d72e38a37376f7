# (A/B script of variants that were measured and NOT kept: the libomok_hip_hg / nk / hgnk builds are not made any more -- profiles/r4p_step_board_heads_variants.txt)
# round 4: k_step_board variants on one box, interleaved: heads_board_one_round vs the generic heads (-DAO_HEADS_GENERIC), with and
# without the kernel-argument touch at the top (-DAO_NO_KTOUCH)
for rep in 1 2 3 4; do
  for tag in "" hg nk hgnk; do
    echo -n "lib '$tag': "; AO_LIB_TAG=$tag python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"
  done
done
