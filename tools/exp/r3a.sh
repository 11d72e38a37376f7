# round 3, first GPU call: (1) subnormal fp16 inputs of the MFMA, (2) activation-bytes knock-outs of k_trunk16hb
hipcc --offload-arch=gfx950 tools/mfma_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
for rep in 1 2 3; do for t in "" ko5 ko6 ko10; do echo "== tag=[$t] rep $rep"; AO_LIB_TAG=$t python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
