# samples the GPU's power / clocks while the trunk runs back to back (evidence for the power-limit reading of the A/B runs)
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from alpha_omok_amd.pvnet import PVNet
torch.manual_seed(0)
net = PVNet(4, 5, 128, 9).eval().to_native(0)
x = (torch.rand(4096, 5, 9, 9, device="cuda") < 0.3).float()
t0 = time.time()
n = 0
while time.time() - t0 < 14:
    for _ in range(200):
        net(x)
    torch.cuda.synchronize()
    n += 200
print("launches", n, "avg ms", (time.time() - t0) / n * 1e3)
PY
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hbm)" | tr -s ' ' | head -8
  echo "--"
  sleep 1
done
wait
echo "== idle"
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | head -3
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
