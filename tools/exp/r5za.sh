# round 5, after the in-place re-rooting commit: traffic / kernel stats re-taken under the new source hash (profile_round), and the
# re-rooting A/B on the trained net (AO_COMPACT_ALWAYS=1 = the copy after every move of rounds 1 - 5), one box
python tools/profile_round.py r5za > gpurun_out/r5za_profile_round.log 2>&1; tail -2 gpurun_out/r5za_profile_round.log
rm -rf gpurun_out/profiles_r5za/raw_*
for rep in 1 2; do
for sw in 0 1; do
  if [ $sw = 1 ]; then export AO_COMPACT_ALWAYS=1; else unset AO_COMPACT_ALWAYS; fi
  echo "## AO_COMPACT_ALWAYS=$sw, trained net (take $rep)"
  python tools/time_move_phases.py --weights profiles/r4_trained_9x9_4block.pt --steps 8 --warm-plies 6 2>&1 | tail -6
done
done > gpurun_out/r5za_reroot_ab.txt 2>&1
unset AO_COMPACT_ALWAYS
echo "## random-init net" >> gpurun_out/r5za_reroot_ab.txt
python tools/time_move_phases.py --steps 8 --warm-plies 6 2>&1 | tail -6 >> gpurun_out/r5za_reroot_ab.txt
AO_COMPACT_ALWAYS=1 python tools/time_move_phases.py --steps 8 --warm-plies 6 2>&1 | tail -6 >> gpurun_out/r5za_reroot_ab.txt
cut -c1-200 gpurun_out/r5za_reroot_ab.txt
