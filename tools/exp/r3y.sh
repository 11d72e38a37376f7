# LDS operand reads of the trunk pinned a whole cell ahead of their use (a sched_barrier after the two ds_reads of the j-loop, built
# for this experiment as the default library) against the scheduler's placement (tag nopin): profiles/r3y_trunk16h_lds_read_distance.txt
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for tag in "" nopin; do
    echo -n "tag=[$tag] 9x9 4096  "; AO_LIB_TAG=$tag python tools/time_net.py 4096 4 9 5 | tail -1
  done
done
for tag in "" nopin; do
  echo -n "tag=[$tag] 9x9 2048 per-layer  "; AO_LIB_TAG=$tag python tools/time_net.py 2048 4 9 6 | tail -1
  echo -n "tag=[$tag] 9x9 512 per-layer  "; AO_LIB_TAG=$tag python tools/time_net.py 512 4 9 6 | tail -1
  echo -n "tag=[$tag] 15x15 1024 10 blocks  "; AO_LIB_TAG=$tag python tools/time_net.py 1024 10 15 5 | tail -1
  echo -n "tag=[$tag] 9x9 4096 10 blocks  "; AO_LIB_TAG=$tag python tools/time_net.py 4096 10 9 5 | tail -1
done
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -2
