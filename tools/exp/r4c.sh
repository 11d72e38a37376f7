# round 4: the cout-pair-split per-layer kernel k_layer16hk against k_layer16h at medium batches (9x9, 4 blocks, whole forward)
python -m pytest tests/test_gpu_net.py -x -q -k "ksplit or full_size_batch or forward_vs_torch" > gpurun_out/r4c_pytest.log 2>&1; tail -3 gpurun_out/r4c_pytest.log
for b in 256 384 512 768 1024 1280 1536 2048 3000 4096; do
  for ks in "0,0" "1,64" "1,128"; do
    echo -n "boards $b AO_KSPLIT=$ks: "; AO_KSPLIT=$ks python tools/time_net.py $b 4 9 0 2>&1 | grep forward
  done
done > gpurun_out/r4c_ksplit.txt
cat gpurun_out/r4c_ksplit.txt
