# round 4: cache policy of the activation STORES in the per-layer path (AO_AUX_ST: 16 = sc1, 17 = sc0 sc1, 2 = nt): do write-through stores shorten the
# gap between consecutive layer launches (the end-of-kernel L2 write-back)? whole forward (tools/time_net.py), interleaved, twice
for rep in 1 2; do for tag in "" st16 st17 st2; do for b in 1024 2048; do
  echo -n "lib '$tag' "; AO_LIB_TAG=$tag python tools/time_net.py $b 4 9 0 2>&1 | grep forward | cut -c1-110
done; done; done
for tag in "" st16 st17 st2; do echo -n "lib '$tag' 15x15 "; AO_LIB_TAG=$tag python tools/time_net.py 1024 10 15 0 2>&1 | grep forward | cut -c1-110; done
for tag in "" st16; do echo -n "lib '$tag' 4096 "; AO_LIB_TAG=$tag python tools/time_net.py 4096 4 9 0 2>&1 | grep forward | cut -c1-110; done
