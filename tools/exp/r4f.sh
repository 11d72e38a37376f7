# round 4: the staging / epilogue knock-outs of round 2 re-taken with DATA-LIKE operands (AO_KO 13 / 14 / 15; 3 / 4 = the round-2 builds,
# whose MFMA operands froze or went to zero) for the resident 9x9 trunk (4096 boards, 4 blocks, fmt 0) and the per-layer 15x15 kernel
# (1024 boards, 10 blocks); package power and shader clock beside each
hipcc --offload-arch=gfx950 -O3 tools/tree_layout_latency.hip -o /tmp/tll 2>/dev/null && /tmp/tll > gpurun_out/r4f_tree_layout_latency.txt 2>&1; cat gpurun_out/r4f_tree_layout_latency.txt
run() {  # $1 boards $2 blocks $3 board
python - "$1" "$2" "$3" <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from alpha_omok_amd.pvnet import PVNet
boards, nb, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
net = PVNet(nb, 5, 128, B).eval().to_native(0)
net.set_mode(5)
x = (torch.rand(boards, 5, B, B, device="cuda") < 0.3).float()
for _ in range(30): net(x)
torch.cuda.synchronize()
net.conv_timing(True)
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(100): net(x)
    torch.cuda.synchronize(); n += 100
ms, cnt = net.conv_timing(False)
print("  forward avg ms %.4f   %s: %.4f ms per launch" % ((time.time() - t0) / n * 1e3, net.dominant_kernel(boards)[0].split(" (")[0], ms / max(cnt, 1)))
PY
sleep 4
for i in 1 2 3; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done
wait
}
for t in "" ko13 ko14 ko15 ko3 ko4; do
  echo "== 9x9 resident trunk, 4096 boards, tag=[$t]"; AO_TRUNK_FMT=0 AO_LIB_TAG=$t run 4096 4 9
done
for t in "" ko13 ko14 ko15 ko3 ko4; do
  echo "== 15x15 per-layer, 1024 boards, 10 blocks, tag=[$t]"; AO_LIB_TAG=$t run 1024 10 15
done
# the reference's default depth (main.py:33 N_BLOCKS = 10), trained by the engine for 30 minutes, and the same evidence on it
python -m pytest tests/test_gpu_net.py -x -q -k "ksplit" > gpurun_out/r4f_pytest.log 2>&1; tail -2 gpurun_out/r4f_pytest.log
python tools/train_omok.py --out gpurun_out/r4f_train10 --minutes 30 --board 9 --blocks 10 --planes 128 --sims 400 \
    --games 2048 --steps 800 --batch 512 --eval-every 10 --eval-dense-until 6 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 > gpurun_out/r4f_train10.log 2>&1
grep '"kind": "elo"' gpurun_out/r4f_train10/log.jsonl | grep -v '"vs": "iter[1-9]' | cut -c1-230 | tail -12
grep '"kind": "iter"' gpurun_out/r4f_train10/log.jsonl | tail -2 | cut -c1-500
python tools/check_trained_net.py --ckpt gpurun_out/r4f_train10/final.pt --blocks 10 --games 1024 --out gpurun_out/r4f_trained_net10.json 2>&1 | grep -v "amdgpu.ids\|WARNING" | tail -10
rm -f gpurun_out/r4f_train10/ckpt_0.pt
