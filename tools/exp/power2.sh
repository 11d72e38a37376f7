# round 3: package power and shader clock while the trunk variants run back to back (4096 boards): is every variant at the cap,
# and does the clock carry the difference?   fmt0 / fmt1 = shipped formats, ko10 = half of every low fragment ZERO in LDS,
# ko12 = the same HBM traffic as ko10 with data-like operands
run() {  # $1 = label, env in front
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from alpha_omok_amd.pvnet import PVNet
torch.manual_seed(0)
net = PVNet(4, 5, 128, 9).eval().to_native(0)
net.set_mode(5)
x = (torch.rand(4096, 5, 9, 9, device="cuda") < 0.3).float()
for _ in range(50): net(x)
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 9:
    for _ in range(200): net(x)
    torch.cuda.synchronize(); n += 200
print("  launches", n, "avg ms %.4f" % ((time.time() - t0) / n * 1e3))
PY
sleep 4.5
for i in 1 2 3; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done
wait
}
echo "== fmt0";  AO_TRUNK_FMT=0 run
echo "== fmt1";  AO_TRUNK_FMT=1 run
echo "== ko10 (fmt0 code, half the low operands zero)";  AO_TRUNK_FMT=0 AO_LIB_TAG=ko10 run
echo "== ko12 (fmt0 code, ko10's traffic, data-like operands)";  AO_TRUNK_FMT=0 AO_LIB_TAG=ko12 run
