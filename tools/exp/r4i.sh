# round 4: what the launches around the trunk convs cost on the per-layer path (conv1, k_head_conv, k_head_fc): rocprofv3 stats of the forward alone
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 4 9" "2048 4 9" "256 4 9" "1024 10 15"; do
  set -- $cfg
  rm -rf /tmp/prof_i; rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o p -- python $GRAFT_REPO_ROOT/tools/time_net.py $1 $2 $3 0 > /dev/null 2>&1
  echo "== forward at $1 boards, $2 blocks, ${3}x${3}"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_i -name "*.db" | head -1) | head -9 | cut -c1-170
done
