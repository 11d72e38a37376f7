#!/bin/bash
# round 6: (1) k_boardh<15> with the ResBlock input not kept (AO_BKO=6, wrong results): the price of its 60 registers / the spill's scratch traffic;
# (2) HBM counters of the 15x15 shape now that the heads' 1x1 convs are taken inside the kernel; (3) fp16-grid learning check (tools/exp/r6e.sh)
R=$GRAFT_REPO_ROOT
{
for i in 1 2; do
python tools/time_net.py 1024 10 15 0 2>/dev/null | sed 's/^/product        : /'
AO_LIB_TAG=bko6 python tools/time_net.py 1024 10 15 0 2>/dev/null | sed "s/^/AO_BKO=6 no res : /"
done
} | tee gpurun_out/r6f_boardh_residual_ko.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --board 15 --games 1024 --sims 40 --blocks 10 --steps 1 --warmup 0 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net"
{
rocprofv3 --kernel-trace --stats -d /tmp/p15_stats -o s -- $B > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p15_stats -name "*.db" | head -1) | head -8
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p15_$c -o p -- $B > /dev/null 2>&1
  echo "## $c (KiB per dispatch)"
  python3 $R/tools/rocpd_summary.py pmc $(find /tmp/p15_$c -name "*.db" | head -1) | grep -E "k_boardh|k_head|kernel" | head -6
done
} | tee $R/gpurun_out/r6f_boardh_15x15_stats_and_traffic.txt
cd $R
bash tools/exp/r6e.sh
