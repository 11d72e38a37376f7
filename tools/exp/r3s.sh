# per-board split-fp16 tile kernel with a board loop (weights once per workgroup) against the per-layer kernels
cd $GRAFT_REPO_ROOT
for b in 8 16 32 64 128 256 512 1024 2048; do
  echo "== $b boards"
  echo -n "mode3 bpw=1   "; AO_BPW=1 python tools/time_net.py $b 4 9 3 | tail -1
  echo -n "mode3 auto    "; python tools/time_net.py $b 4 9 3 | tail -1
  echo -n "mode3 bpw*2   "; AO_BPW=$(( (b*24/768)*2 > 0 ? (b*24/768)*2 : 1 )) python tools/time_net.py $b 4 9 3 | tail -1
  echo -n "mode6 layers  "; python tools/time_net.py $b 4 9 6 | tail -1
done
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -2
