# round 3: (1) is the traffic knock-out's -15 % real? ko10 leaves half of every low fragment ZERO in LDS, ko12 keeps it data-like
# (2) HBM bytes per launch of the 4-byte and the 3-byte trunk (rocprofv3 PMC, separate passes)
for rep in 1 2; do for t in "" ko10 ko12; do echo "== tag=[$t] rep $rep"; AO_LIB_TAG=$t python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
cd /tmp; export TMPDIR=/tmp
for f in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  d=$GRAFT_REPO_ROOT/gpurun_out/r3d_pmc_${f}_$c
  AO_TRUNK_FMT=$f rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python $GRAFT_REPO_ROOT/tools/time_net.py 4096 4 9 5 > /dev/null 2>&1
  echo "== AO_TRUNK_FMT=$f $c"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $(find $d -name "*.db" | head -1) | grep -i trunk
  rm -rf $d
done; done
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_net.py -x -q -k "fp16_range" 2>&1 | tail -3
