# round 4: networks of 160 .. 256 planes on the fp32 layer kernels (two workgroups per group and row chunk); net + drop-in tests, timing
python -m pytest tests/test_gpu_net.py tests/test_gpu_dropin.py -x -q > gpurun_out/r4j_pytest.log 2>&1; tail -3 gpurun_out/r4j_pytest.log
python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from alpha_omok_amd.pvnet import PVNet
for planes, boards in ((128, 1024), (192, 1024), (256, 1024), (256, 4096), (256, 64)):
    torch.manual_seed(0)
    net = PVNet(4, 5, planes, 9).eval().to_native(0)
    if planes == 128: net.set_mode(4)
    x = (torch.rand(boards, 5, 9, 9, device="cuda") < 0.3).float()
    for _ in range(3): net(x)
    torch.cuda.synchronize(); net.conv_timing(True); t0 = time.perf_counter()
    for _ in range(10): net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ms, cnt = net.conv_timing(False); name, flop = net.dominant_kernel(boards)
    print("%d planes, %d boards: forward %.3f ms; %s %.3f ms per conv = %.1f TFLOP/s (fp32 MFMA peak 157)" % (planes, boards, dt * 1e3, name.split(" (")[0], ms / max(cnt, 1), flop / (ms / max(cnt, 1) * 1e-3) / 1e12))
PY
