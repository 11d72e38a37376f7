# round 4: k_row16hk (one workgroup per group x output row x cout pair) for small batches: correctness, then the forward by batch size with and without
timeout 300 python tools/exp/r4v_check.py 2>&1 | grep -v amdgpu
for b in 48 64 96 128 192 256 384 512 640 768 1024; do
  for rk in "0,-1" "1,4096"; do
    echo -n "boards $b AO_ROWK=$rk: "; AO_ROWK=$rk python tools/time_net.py $b 4 9 0 2>&1 | grep forward | cut -c1-120
  done
done
