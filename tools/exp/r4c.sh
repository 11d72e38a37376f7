# round 4: the cout-pair-split per-layer kernel k_layer16hk against k_layer16h at medium batches (9x9, 4 blocks, whole forward)
python -m pytest tests/test_gpu_net.py -x -q -k "ksplit or full_size_batch or forward_vs_torch" > gpurun_out/r4c_pytest.log 2>&1; tail -3 gpurun_out/r4c_pytest.log
for b in 256 384 512 768 1024 1280 1536 2048 3000 4096; do
  for ks in "0,0" "1,64" "1,128"; do
    echo -n "boards $b AO_KSPLIT=$ks: "; AO_KSPLIT=$ks python tools/time_net.py $b 4 9 0 2>&1 | grep forward
  done
done > gpurun_out/r4c_ksplit.txt
cat gpurun_out/r4c_ksplit.txt
# single game: where the tree step's cycles go (-DAO_PROF build, unfused step so that k_expand_select's own phase ticks print)
for i in 1 2; do AO_LIB_TAG=prof AO_PROF_TREE=1 AO_FUSED_STEP=0 python tools/time_single_game.py --moves 2 2>&1 | grep "AO_PROF k_expand" | tail -2; done > gpurun_out/r4c_tree_phases.txt
AO_LIB_TAG=prof AO_PROF_TREE=1 python tools/time_single_game.py --moves 2 2>&1 | grep "AO_PROF" | tail -4 >> gpurun_out/r4c_tree_phases.txt
cat gpurun_out/r4c_tree_phases.txt
# the steep part of the learning curve: evaluation after EVERY iteration (vs iteration 0 and vs the PUCT rollout agent at 400 playouts)
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_tree_parity.py -x -q > gpurun_out/r4c_pytest2.log 2>&1; tail -3 gpurun_out/r4c_pytest2.log
python tools/train_omok.py --out gpurun_out/r4c_curve --minutes 6 --iters 12 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 100 --eval-dense-until 12 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 > gpurun_out/r4c_curve.log 2>&1
grep '"kind": "elo"' gpurun_out/r4c_curve/log.jsonl | cut -c1-250
