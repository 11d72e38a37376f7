# (A/B script of a variant that was measured and NOT kept: k_layer16h2 / AO_XT=2,3 are not in the tree -- profiles/r4y_layer16h_15x15_narrow_tiles.txt)
# round 4: 15x15 per-layer path with NARROW column tiles and two workgroups per CU (k_layer16h2, AO_XT=2 / 3) against XT=4 (256 workgroups, one per CU):
# correctness against torch fp32 first, then the forward (1024 boards, 10 blocks) and the bench of configs[4]'s per-GPU shape
python - <<'PY'
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from alpha_omok_amd.pvnet import PVNet
import pvnet_weights
for B, nb, boards in ((15, 10, 1024), (15, 2, 100), (13, 3, 300), (11, 2, 777)):
    ref = PVNet(nb, 5, 128, B); ref.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(nb, 5, 128, B, 7).items()}); ref.eval()
    rs = np.random.RandomState(B); x = (rs.rand(boards, 5, B, B) < 0.3).astype(np.float32)
    idx = rs.choice(boards, 48, replace=False)
    with torch.no_grad(): rp, rv = ref(torch.from_numpy(x[idx]))
    outs = {}
    for xt in ("4", "2", "3"):
        os.environ["AO_XT"] = xt
        net = ref.to_native(0)
        p, v = net(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
        outs[xt] = (p.cpu().numpy(), v.cpu().numpy())
        e = max(np.abs(outs[xt][0][idx] - rp.numpy()).max(), np.abs(outs[xt][1][idx] - rv.numpy()).max())
        print("B %d blocks %d boards %d XT=%s: vs torch fp32 %.2e  == XT=4 bit for bit: %s  status %d" % (B, nb, boards, xt, e, np.array_equal(outs[xt][0], outs["4"][0]) and np.array_equal(outs[xt][1], outs["4"][1]), net.status()))
        net.close()
PY
for rep in 1 2; do for xt in 4 2 3; do echo -n "AO_XT=$xt "; AO_XT=$xt python tools/time_net.py 1024 10 15 0 2>&1 | grep forward | cut -c1-120; done; done
for xt in 4 2; do
  AO_XT=$xt python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net > gpurun_out/r4y_b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4y_b.json')); print('AO_XT=$xt bench 15x15: value %.1f  conv %.4f ms  frac %.4f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
