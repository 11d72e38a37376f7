# rocprofv3 stats + HBM counters of the configs[4] per-GPU shape (15x15, 10 blocks, 1024 games, 800 sims): 40 simulations are enough
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --board 15 --games 1024 --sims 40 --blocks 10 --steps 1 --warmup 0 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block"
rocprofv3 --kernel-trace --stats -d /tmp/p15_stats -o s -- $B > /dev/null 2>&1
python3 $R/tools/rocpd_summary.py stats $(find /tmp/p15_stats -name "*.db" | head -1) | head -8
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p15_$c -o p -- $B > /dev/null 2>&1
  echo "## $c (KiB per dispatch)"
  python3 $R/tools/rocpd_summary.py pmc $(find /tmp/p15_$c -name "*.db" | head -1) | grep -E "k_layer16h|k_expand_select|kernel" | head -6
done
