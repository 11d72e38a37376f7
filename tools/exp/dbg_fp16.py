import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, warnings
import pvnet_weights
from alpha_omok_amd.engine import Engine, Net
B, batch, S = 9, 64, 6
sd = pvnet_weights.make_state_dict(2, 5, 128, B, 4)
big = dict(sd); big["bn1.weight"] = (sd["bn1.weight"] * 3.0e5).astype(np.float32)
net = Net(2, 5, 128, B, 0); net.load_state_dict(big); net.set_mode(0)
eng = Engine(B, S, 5, games=batch, noise=True)
eng.seed_all(np.arange(batch, dtype=np.uint32) + 50)
warnings.simplefilter("always")
for ply in range(4):
    pi, vis, pol = eng.search(net, tau=1)
    print("ply", ply, "events", eng.fp16_range_events(), "mode", net._L.ao_net_get_mode(net._h), "status", net.status(clear=False), "vis sum", vis.sum(axis=1)[:4])
    eng.play()
