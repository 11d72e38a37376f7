# (A/B script of a variant that was measured and NOT kept: AO_ROWK_HALVES is not in the tree -- profiles/r4x_row_kernel_column_halves.txt)
# round 4: k_row16hk with the row's cells on TWO workgroups (AO_ROWK_HALVES=2: 48 KB of LDS, three workgroups per CU) against one (72 KB, two per CU):
# correctness (same bits as k_layer16hk), the forward by batch size, extended group ranges
AO_ROWK_HALVES=2 timeout 300 python tools/exp/r4v_check.py 2>&1 | grep -v amdgpu | cut -c1-110
for b in 64 128 256 384 512 640 768 1024 1536 2048; do
  for cfg in "1 1,4096" "2 1,4096"; do
    set -- $cfg
    echo -n "boards $b AO_ROWK_HALVES=$1 AO_ROWK=$2: "; AO_ROWK_HALVES=$1 AO_ROWK=$2 AO_PERBOARD_CELLS=2592 python tools/time_net.py $b 4 9 0 2>&1 | grep forward | cut -c1-120
  done
done
