"""On the GPU box: error of the resident split-fp16 trunk with 4-byte (AO_TRUNK_FMT=0) and 3-byte (default) activations against
a float64 evaluation of the same network (torch CPU), golden-vector networks (tests/pvnet_weights.py) and the bench's
default-initialised one.   python tools/check_trunk_fmt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import pvnet_weights
from alpha_omok_amd.pvnet import PVNet

BATCH, SAMPLE = 3072, 48


def run(ref, x, fmt):
    os.environ["AO_TRUNK_FMT"] = str(fmt)
    net = ref.to_native(0)
    del os.environ["AO_TRUNK_FMT"]
    net.set_mode(5)
    p, v = net(x.cuda())
    torch.cuda.synchronize()
    name = net.dominant_kernel(x.shape[0])[0].split(" (")[0]
    st = net.status()
    net.close()
    return p.cpu().double(), v.cpu().double(), name, st


print("%-34s %-22s %12s %12s" % ("network", "kernel", "max |dp|", "max |dv|"))
for nb, B, seed in ((4, 9, 77), (4, 9, 3), (10, 9, 5), (4, 9, None), (10, 9, None), (4, 7, 11)):
    ref = PVNet(nb, 5, 128, B)
    if seed is None:
        torch.manual_seed(0)
        ref = PVNet(nb, 5, 128, B)
    else:
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(nb, 5, 128, B, seed).items()})
    ref.eval()
    rs = np.random.RandomState(nb + B)
    x = torch.from_numpy((rs.rand(BATCH, 5, B, B) < 0.3).astype(np.float32))
    idx = rs.choice(BATCH, SAMPLE, replace=False)
    with torch.no_grad():
        p64, v64 = ref.double()(x[idx].double())
    ref.float()
    for fmt in (0, 1):
        p, v, name, st = run(ref, x, fmt)
        print("%-34s %-22s %12.2e %12.2e%s" % ("%d blocks, %dx%d, %s" % (nb, B, B, "seed %d" % seed if seed is not None else "default init"),
                                                name, (p[idx] - p64).abs().max().item(), (v[idx] - v64).abs().max().item(),
                                                "  (fp16-range flag set)" if st else ""))
