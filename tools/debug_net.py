import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import pvnet_weights
from alpha_omok_amd.engine import Net

def run(nb, B, planes, batch, seed=1):
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, seed)
    rs = np.random.RandomState(batch)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    outs = []
    for mode in (1, 2):
        net = Net(nb, 5, planes, B, 0); net.load_state_dict(sd); net.set_mode(mode)
        p, v = net(xt); torch.cuda.synchronize()
        outs.append((p.cpu().numpy(), v.cpu().numpy())); net.close()
    dp = np.abs(outs[0][0] - outs[1][0]).max(axis=1)
    bad = np.nonzero(dp > 1e-4)[0]
    print("nb %d B %d planes %d batch %d: max dp %.3g dv %.3g bad boards %s" % (nb, B, planes, batch, dp.max(), np.abs(outs[0][1]-outs[1][1]).max(), bad[:40].tolist()))

for cfg in [(0,9,128,16),(1,9,128,17),(1,15,128,20),(1,3,32,40)]:
    run(*cfg)
