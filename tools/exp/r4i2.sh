# round 4, end: rocprofv3 stats of the forward alone at the batch sizes k_row16hk serves (64 .. 640 boards) -- the trunk conv against the launches around it
cd /tmp && export TMPDIR=/tmp
for cfg in "64 4 9" "128 4 9" "256 4 9" "512 4 9" "640 4 9"; do
  set -- $cfg
  rm -rf /tmp/prof_i; rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o p -- python $GRAFT_REPO_ROOT/tools/time_net.py $1 $2 $3 0 > /dev/null 2>&1
  echo "== forward at $1 boards, $2 blocks, ${3}x${3}"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_i -name "*.db" | head -1) | head -7 | cut -c1-170
done
