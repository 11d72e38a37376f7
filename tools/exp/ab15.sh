echo "== default plan"; python tools/time_net.py 1024 10 15 0 2>&1 | tail -1
for xt in 5 4; do for nch in 1 2 3 4 5; do echo "== XT=$xt NCH=$nch"; AO_XT=$xt AO_NCH=$nch python tools/time_net.py 1024 10 15 0 2>&1 | tail -1; done; done
for b in 256 512 2048 4096; do echo "== boards=$b default"; python tools/time_net.py $b 10 15 0 2>&1 | tail -1; for xt in 5 4; do echo "== boards=$b XT=$xt (nch auto)"; AO_XT=$xt python tools/time_net.py $b 10 15 0 2>&1 | tail -1; done; done
AO_XT=4 python -m pytest tests/test_gpu_net.py -x -q -k "15" 2>&1 | tail -3
AO_XT=4 AO_NCH=1 python -m pytest tests/test_gpu_net.py -x -q -k "15" 2>&1 | tail -3
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -3
