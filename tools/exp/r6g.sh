#!/bin/bash
# round 6: with two products and a trained network's activations the resident trunk runs at 2.16 of 2.4 GHz with the MFMA pipe busy 0.60 of the
# cycles -- less power-limited than the three-product random-init headline (1.9 GHz, 0.70). Does removing row synchronisation pay NOW?
# sb = -DAO_SPLIT_BARRIER=1 (correct results), ko9 = no row barrier at all (wrong results: the upper bound)
W="--weights profiles/r4_trained_9x9_4block.pt"
for i in 1 2; do
for tag in "" sb ko9; do
  AO_LIB_TAG=$tag python tools/time_net.py 4096 4 9 0 $W --fp16-grid 2>/dev/null | sed "s/^/[${tag:-product}] /"
  AO_LIB_TAG=$tag python tools/time_net.py 4096 4 9 0 $W 2>/dev/null | sed "s/^/[${tag:-product}] /"
done
done | tee gpurun_out/r6g_trunk_w16_row_sync_ab.txt
