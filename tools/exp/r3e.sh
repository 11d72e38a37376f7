# round 3: do fewer significant bits in the weights' low halves (wl) lower the MFMA operand activity enough to matter?
for b in 11 8 6; do echo "== AO_WL_BITS=$b"; AO_WL_BITS=$b python tools/check_trunk_fmt.py 2>&1 | grep -E "seed 77|seed 5|default init" | grep "4, 1>\|10 blocks"; done
for rep in 1 2 3; do for b in 11 8 6; do echo "== AO_WL_BITS=$b rep $rep"; AO_WL_BITS=$b python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
