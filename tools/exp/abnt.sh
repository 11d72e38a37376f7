for rep in 1 2; do for t in "" stage2 res2 st2 all2 ldnt st16 stage16; do echo "== tag=[$t] rep $rep"; AO_LIB_TAG=$t python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
