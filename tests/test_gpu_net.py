"""-m gpu: the hand-written fp32 MFMA PVNet forward against (1) golden (p, v) produced by the
reference's model.PVNet and (2) a torch fp32 reference, tolerance 1e-4 absolute on p and v
(BASELINE.json north_star); then the fused search (ao_search) against the oracle."""
import numpy as np
import pytest

import pvnet_weights
from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _native(nb, C, planes, B, seed):
    from alpha_omok_amd.engine import Net
    net = Net(nb, C, planes, B, 0)
    net.load_state_dict(pvnet_weights.make_state_dict(nb, C, planes, B, seed))
    return net


def _h16_ok(nb, B, planes):
    """mode 5 (split-fp16 MFMA trunk; resident kernel for >= 192 groups of boards up to 9x9, one launch per
    layer otherwise) is built for 128 planes and >= 1 ResBlock"""
    return planes == 128 and nb >= 1


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_golden_gv7_forward(mode):
    """mode 1: one kernel per conv (groups of 32 boards); mode 2: group-resident trunk (16); 5: split-fp16 trunk."""
    import torch
    g = load_golden("gv7_pvnet_forward")
    ran = 0
    for i in range(int(g["count"])):
        nb, B, planes, wseed = g["cfg%d" % i].tolist()
        if mode == 5 and not _h16_ok(nb, B, planes):
            continue
        ran += 1
        net = _native(nb, 5, planes, B, wseed)
        net.set_mode(mode)
        x = torch.from_numpy(g["x%d" % i]).cuda()
        p, v = net(x)
        torch.cuda.synchronize()
        dp = np.abs(p.cpu().numpy() - g["p%d" % i]).max()
        dv = np.abs(v.cpu().numpy() - g["v%d" % i]).max()
        assert dp < TOL and dv < TOL, (nb, B, planes, dp, dv)
        assert abs(p.sum(dim=1).cpu().numpy() - 1).max() < 1e-5
        net.close()
    if mode == 5 and ran == 0:
        pytest.skip("no 128-plane / <= 9x9 case in the fixture")


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("nb,B,planes,batch", [(4, 9, 128, 70), (10, 9, 128, 33), (2, 15, 128, 40),
                                               (3, 9, 64, 32), (1, 3, 32, 5), (2, 7, 96, 64), (2, 7, 128, 20),
                                               (1, 3, 128, 3)])
def test_forward_vs_torch_fp32(nb, B, planes, batch, mode):
    import torch
    from alpha_omok_amd.pvnet import PVNet
    if mode == 5 and not _h16_ok(nb, B, planes):
        pytest.skip("split-fp16 trunk: 128 planes only")
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, 100 + nb)
    ref = PVNet(nb, 5, planes, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref.eval()
    rs = np.random.RandomState(batch)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    x[:, 4] = (rs.rand(batch, 1, 1) < 0.5).astype(np.float32)
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x))
    net = ref.to_native(0)
    net.set_mode(mode)
    p, v = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    dp = np.abs(p.cpu().numpy() - rp.numpy()).max()
    dv = np.abs(v.cpu().numpy() - rv.numpy()).max()
    assert dp < TOL and dv < TOL, (dp, dv)
    net.close()


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_fused_search_matches_stepwise_and_oracle(oracle, mode):
    """ao_search (select -> native PVNet -> expand on one stream) == the stepwise protocol fed by
    the same network through ao_net_forward, and == the oracle replaying those (p, v)."""
    import torch
    from alpha_omok_amd.engine import Engine
    from gpu_helpers import HostEvalRunner
    B, S, G, plies = 9, 64, 33, 3
    net = _native(2, 5, 64, B, 42)
    net.set_mode(mode)
    seeds = [500 + g for g in range(G)]
    e1 = Engine(B, S, 5, games=G, noise=True)
    e2 = Engine(B, S, 5, games=G, noise=True)
    e1.seed_all(seeds)
    e2.seed_all(seeds)
    run = HostEvalRunner(e2)
    rec = [[] for _ in range(G)]
    for t in range(plies):
        tau = np.full(G, 1 if t < 2 else 0, np.int8)
        pi1, vis1, pol1 = e1.search(net, tau=tau)
        # stepwise: evaluate the NCHW planes with the same network, record what each game saw
        e2.begin_move()
        while e2.sims_left() > 0:
            e2.collect_leaves(run.planes.data_ptr())
            e2.sync()
            p, v = net(run.planes)
            torch.cuda.synchronize()
            hp, hv = p.cpu().numpy(), v.cpu().numpy()
            for g in range(G):
                rec[g].append((hp[g].copy(), hv[g].copy()))
            e2.apply_evals(p.data_ptr(), v.data_ptr())
        pi2, vis2, pol2 = e2.end_move(tau)
        np.testing.assert_array_equal(vis1, vis2)
        np.testing.assert_array_equal(pol1, pol2)
        np.testing.assert_array_equal(pi1, pi2)
        a1, w1 = e1.play()
        a2, w2 = e2.play()
        np.testing.assert_array_equal(a1, a2)
        np.testing.assert_array_equal(w1, w2)
        if t == 0:
            first_vis, first_act = vis1.copy(), a1.copy()
    # oracle replay of game 0 and game G-1, first ply (401 recorded evaluations each)
    for g in (0, G - 1):
        cur = [0]

        def replay(moves, planes, sim, g=g, cur=cur):
            i = cur[0]
            cur[0] += 1
            return rec[g][i]

        ag = oracle.Agent(B, S, 5, noise=True, evaluator=replay)
        ag.seed(seeds[g])
        opi, ovis, opol = ag.get_pi((0,), 1)
        np.testing.assert_array_equal(first_vis[g], ovis)
        assert first_act[g] == ag.rng.choice_p(opi)
    e1.close()
    e2.close()
    net.close()


def test_full_size_batch_all_paths_agree():
    """4096 boards (one workgroup per CU in the group-resident trunk): the three trunk paths are
    independent implementations and must agree; a sample is checked against the torch fp32 net.
    Repeated a few times: a stale-cache or race bug in the in-kernel layer hand-off would show up
    as run-to-run differences under full load."""
    import torch
    from alpha_omok_amd.pvnet import PVNet
    nb, B, planes, batch = 4, 9, 128, 4096
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, 321)
    ref = PVNet(nb, 5, planes, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref.eval()
    rs = np.random.RandomState(1)
    x = (rs.rand(batch, 5, B, B) < 0.25).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    net = ref.to_native(0)
    outs = {}
    for mode in (2, 1, 3, 4, 5, 2, 5, 2, 5):
        net.set_mode(mode)
        p, v = net(xt)
        torch.cuda.synchronize()
        p, v = p.cpu().numpy(), v.cpu().numpy()
        assert np.isfinite(p).all() and np.isfinite(v).all()
        if mode in outs:
            np.testing.assert_array_equal(outs[mode][0], p)      # same path twice: bit-identical
            np.testing.assert_array_equal(outs[mode][1], v)
        outs[mode] = (p, v)
    for m in (1, 3, 4, 5):
        # (mode 5 at 4 blocks keeps its activations in the 3-byte format, 19 significand bits: up to ~2e-5 against fp64
        # on these golden-vector weights, profiles/r3a_trunk16h_bytes_ko.txt)
        assert np.abs(outs[2][0] - outs[m][0]).max() < (5e-5 if m == 5 else 2e-5), m
        assert np.abs(outs[2][1] - outs[m][1]).max() < (5e-5 if m == 5 else 2e-5), m
    idx = rs.choice(batch, 96, replace=False)
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x[idx]))
    for m in (2, 5):
        assert np.abs(outs[m][0][idx] - rp.numpy()).max() < TOL, m
        assert np.abs(outs[m][1][idx] - rv.numpy()).max() < TOL, m
    net.close()


@pytest.mark.parametrize("nb,B,batch,seed", [(4, 9, 1024, 321), (4, 9, 1000, 77), (2, 9, 800, 5), (10, 9, 960, 9), (3, 7, 1024, 12),
                                             (2, 5, 1024, 3), (1, 4, 1000, 8), (2, 8, 911, 6), (2, 6, 768, 4),
                                             (4, 9, 2048, 21), (4, 9, 1040, 22), (2, 9, 1999, 23), (10, 9, 1536, 24), (3, 7, 1300, 25),
                                             (2, 5, 2048, 26), (1, 4, 1700, 27), (2, 8, 1111, 28)])
def test_medium_batch_ksplit_layer_kernel(nb, B, batch, seed, monkeypatch):
    """Medium batches (48 .. 64 groups of 16 boards, boards up to 9x9) run their trunk convs as k_layer16hk: a group split
    over four workgroups by cout pairs, the eight waves of a workgroup splitting the contraction by input block, partial tiles
    exchanged through LDS (net_layer_ksplit.hpp). The two-workgroup form (cout quads, 65 .. 128 groups: correct, not faster,
    not planned by default) is exercised through AO_KSPLIT. Checked against the per-layer kernel it replaces (mode 6), the
    fp32-MFMA kernels (mode 4: no fp16 anywhere) and torch fp32; twice, bit-identical (the exchange adds in a fixed order);
    ragged last group (batch not a multiple of 16), residual and non-residual layers, in-place second conv of a block."""
    import torch
    from alpha_omok_amd.pvnet import PVNet
    sd = pvnet_weights.make_state_dict(nb, 5, 128, B, seed)
    ref = PVNet(nb, 5, 128, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref.eval()
    rs = np.random.RandomState(batch + B)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    x[:, 4] = (rs.rand(batch, 1, 1) < 0.5).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    monkeypatch.setenv("AO_KSPLIT", "48,64,128")
    net = ref.to_native(0)
    monkeypatch.delenv("AO_KSPLIT")
    outs = {}
    for mode in (5, 6, 4, 5):
        net.set_mode(mode)
        p, v = net(xt)
        torch.cuda.synchronize()
        if mode == 5:
            assert net.dominant_kernel(batch)[0].startswith("k_layer16hk<%d, %d>" % (B, 4 if batch <= 1024 else 2)), net.dominant_kernel(batch)[0]
        p, v = p.cpu().numpy(), v.cpu().numpy()
        assert np.isfinite(p).all() and np.isfinite(v).all() and net.status() == 0
        if mode in outs:
            np.testing.assert_array_equal(outs[mode][0], p)
            np.testing.assert_array_equal(outs[mode][1], v)
        outs[mode] = (p, v)
    for m in (6, 4):       # (two fp32-equivalent evaluations of a 10-block golden-vector network differ by up to 1e-4: the
        # tower of synthetic weights amplifies fp32 rounding; the bar is the comparison with torch fp32 below)
        assert np.abs(outs[5][0] - outs[m][0]).max() < (1.5e-4 if nb > 6 else 2e-5), m
        assert np.abs(outs[5][1] - outs[m][1]).max() < (1.5e-4 if nb > 6 else 2e-5), m
    idx = rs.choice(batch, 128, replace=False)
    idx[:4] = [0, 15, batch - 1, batch - (batch % 16 or 16)]        # first group, last (ragged) group
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x[idx]))
    assert np.abs(outs[5][0][idx] - rp.numpy()).max() < TOL and np.abs(outs[5][1][idx] - rv.numpy()).max() < TOL
    net.close()


@pytest.mark.parametrize("nb,B,batch,seed", [(4, 9, 128, 1), (4, 9, 100, 2), (10, 9, 333, 3), (2, 9, 750, 4), (3, 7, 200, 5), (2, 5, 512, 6),
                                             (1, 4, 200, 7), (2, 8, 48, 8), (2, 6, 640, 9), (4, 9, 33, 10)])
def test_small_batch_row_kernel(nb, B, batch, seed, monkeypatch):
    """Small and medium-small batches (from 33 boards of 9x9 up to 47 groups of 16) run their trunk convs as k_row16hk: one
    workgroup per (group, output row, cout pair), two per CU, the waves splitting the contraction as in k_layer16hk
    (net_layer_ksplit.hpp). Same arithmetic in the same order as k_layer16hk: BIT-identical to a network whose every group count
    is planned as k_layer16hk; against the per-layer kernel (mode 6), the fp32-MFMA kernels (mode 4) and torch fp32; twice,
    bit-identical; ragged last group; and a 400-free search through it (ao_search: tree kernels + bit planes) against the
    step-wise protocol with the same network."""
    import torch
    from alpha_omok_amd.engine import Engine
    from alpha_omok_amd.pvnet import PVNet
    sd = pvnet_weights.make_state_dict(nb, 5, 128, B, seed)
    ref = PVNet(nb, 5, 128, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref.eval()
    rs = np.random.RandomState(batch + B)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    x[:, 4] = (rs.rand(batch, 1, 1) < 0.5).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    net = ref.to_native(0)
    monkeypatch.setenv("AO_ROWK", "0,-1")
    monkeypatch.setenv("AO_KSPLIT", "1,4096,0")
    net_k = ref.to_native(0)
    monkeypatch.delenv("AO_ROWK")
    monkeypatch.delenv("AO_KSPLIT")
    assert net.dominant_kernel(batch)[0].startswith("k_row16hk<%d>" % B), net.dominant_kernel(batch)[0]
    assert net_k.dominant_kernel(batch)[0].startswith("k_layer16hk<%d, 4>" % B), net_k.dominant_kernel(batch)[0]
    outs = {}
    for mode in (5, 6, 4, 5):
        net.set_mode(mode)
        p, v = net(xt)
        torch.cuda.synchronize()
        p, v = p.cpu().numpy(), v.cpu().numpy()
        assert np.isfinite(p).all() and np.isfinite(v).all() and net.status() == 0
        if mode in outs:
            np.testing.assert_array_equal(outs[mode][0], p)
            np.testing.assert_array_equal(outs[mode][1], v)
        outs[mode] = (p, v)
    pk, vk = net_k(xt)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(outs[5][0], pk.cpu().numpy())
    np.testing.assert_array_equal(outs[5][1], vk.cpu().numpy())
    for m in (6, 4):
        assert np.abs(outs[5][0] - outs[m][0]).max() < (1.5e-4 if nb > 6 else 2e-5), m
        assert np.abs(outs[5][1] - outs[m][1]).max() < (1.5e-4 if nb > 6 else 2e-5), m
    idx = rs.choice(batch, min(batch, 96), replace=False)
    idx[:4] = [0, 15, batch - 1, batch - (batch % 16 or 16)]
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x[idx]))
    assert np.abs(outs[5][0][idx] - rp.numpy()).max() < TOL and np.abs(outs[5][1][idx] - rv.numpy()).max() < TOL
    net_k.close()
    if B >= 5 and batch <= 200:
        # a search with that many games: the fused loop (bit planes -> conv1 -> k_row16hk x 2 nb -> head kernels -> k_expand_select)
        # against the step-wise protocol with the same network
        net.set_mode(0)
        S, G = 10, batch
        a, b2 = Engine(B, S, 5, games=G, noise=True), Engine(B, S, 5, games=G, noise=True)
        seeds = np.arange(G, dtype=np.uint32) + 3
        a.seed_all(seeds); b2.seed_all(seeds)
        pi, vis, pol = a.search(net, tau=1)
        planes_t = torch.zeros((G, 5, B, B), dtype=torch.float32, device="cuda")
        b2.begin_move()
        while b2.sims_left() > 0:
            b2.collect_leaves(planes_t.data_ptr()); b2.sync()
            pp, vv = net(planes_t); torch.cuda.synchronize()
            b2.apply_evals(pp.data_ptr(), vv.data_ptr())
        pi2, vis2, pol2 = b2.end_move(np.ones(G, np.int8))
        np.testing.assert_array_equal(vis, vis2)
        np.testing.assert_array_equal(pol, pol2)
        a.close(); b2.close()
    net.close()


@pytest.mark.parametrize("nb,B,planes,batch", [(2, 9, 256, 40), (1, 5, 192, 700), (2, 15, 160, 24), (3, 9, 224, 1024), (1, 3, 256, 5),
                                               (2, 9, 288, 40), (1, 9, 512, 600), (2, 15, 384, 20), (1, 7, 320, 1), (1, 3, 512, 3)])
def test_wide_networks_run_on_the_fp32_layer_kernels(nb, B, planes, batch):
    """model.PVNet takes any `planes` (model.py:76-85). 160 .. 512 planes (multiples of 32; 256 until round 6) run natively on the
    row-chunked fp32-MFMA layer kernels for every batch size -- a group's output channels beyond 128 go to a second, third, fourth
    workgroup of k_layer16 -- whatever mode is asked for; checked against torch fp32, and through a fused search against the
    stepwise one."""
    import torch
    from alpha_omok_amd.engine import Engine
    from alpha_omok_amd.pvnet import PVNet
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, 40 + planes)
    ref = PVNet(nb, 5, planes, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref.eval()
    rs = np.random.RandomState(batch)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x))
    net = ref.to_native(0)
    for mode in (0, 3, 2, 1):
        net.set_mode(mode)
        assert net.dominant_kernel(batch)[0].startswith("k_layer16<%d>" % B)
        p, v = net(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert np.abs(p.cpu().numpy() - rp.numpy()).max() < TOL and np.abs(v.cpu().numpy() - rv.numpy()).max() < TOL, mode
    if batch <= 64 and B >= 5:
        S, G = 12, min(batch, 8)
        a, b2 = Engine(B, S, 5, games=G, noise=True), Engine(B, S, 5, games=G, noise=True)
        seeds = np.arange(G, dtype=np.uint32) + 3
        a.seed_all(seeds); b2.seed_all(seeds)
        if planes > 256:
            # ... and through the evaluator a drop-in caller goes through (agents.py:171-178): the module is exported, not called
            import warnings
            from alpha_omok_amd.evaluator import Evaluator
            ev = Evaluator(0)
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                assert ev.native_net(ref, B, 5) is not None
                pi, vis, pol = ev.search(a, ref, 1)
        else:
            pi, vis, pol = a.search(net, tau=1)
        planes_t = torch.zeros((G, 5, B, B), dtype=torch.float32, device="cuda")
        b2.begin_move()
        while b2.sims_left() > 0:
            b2.collect_leaves(planes_t.data_ptr()); b2.sync()
            pp, vv = net(planes_t); torch.cuda.synchronize()
            b2.apply_evals(pp.data_ptr(), vv.data_ptr())
        pi2, vis2, pol2 = b2.end_move(np.ones(G, np.int8))
        np.testing.assert_array_equal(vis, vis2)
        np.testing.assert_array_equal(pol, pol2)
        a.close(); b2.close()
    net.close()


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("nb,B,seed", [(4, 9, 77), (10, 9, 5), (6, 7, 12), (1, 5, 3)])
def test_resident_trunk_both_activation_formats(nb, B, seed, fmt, monkeypatch):
    """The resident split-fp16 trunk keeps its activations between layers as two fp16 halves (4 bytes, AO_TRUNK_FMT=0) or as
    an fp16 high half + one low byte (3 bytes, 19 significand bits, AO_TRUNK_FMT=1, opt-in since round 4): both against the
    torch fp32 network on golden-vector weights at 3072 boards (192 groups: the resident kernel); the default is the 4-byte
    format at every depth."""
    import torch
    from alpha_omok_amd.pvnet import PVNet
    batch = 3072
    ref = PVNet(nb, 5, 128, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(nb, 5, 128, B, seed).items()})
    ref.eval()
    rs = np.random.RandomState(seed)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    idx = rs.choice(batch, 48, replace=False)
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x[idx]))
    monkeypatch.setenv("AO_TRUNK_FMT", str(fmt))
    net = ref.to_native(0)
    monkeypatch.delenv("AO_TRUNK_FMT")
    net.set_mode(5)
    p, v = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert net.dominant_kernel(batch)[0].startswith("k_trunk16h<%d, 4, %d>" % (B, fmt))
    assert net.status() == 0
    assert np.abs(p.cpu().numpy()[idx] - rp.numpy()).max() < TOL and np.abs(v.cpu().numpy()[idx] - rv.numpy()).max() < TOL
    assert abs(p.sum(dim=1).cpu().numpy() - 1).max() < 1e-5
    net.close()
    auto = ref.to_native(0)
    auto.set_mode(5)
    assert auto.dominant_kernel(batch)[0].startswith("k_trunk16h<%d, 4, 0>" % B)
    auto.close()


@pytest.mark.parametrize("B,batch", [(9, 4096), (9, 40), (15, 24)])
def test_split_fp16_forward_takes_arbitrary_float_planes(B, batch):
    """ao_net_forward's input is any float32 planes, not only the engine's 0/1 planes: conv1 of the split-fp16
    kernels (resident for 4096 boards, per layer otherwise) splits its input like every other layer."""
    import torch
    from alpha_omok_amd.pvnet import PVNet
    torch.manual_seed(B + batch)
    ref = PVNet(2, 5, 128, B).eval()
    x = torch.randn(batch, 5, B, B) * 3.0
    net = ref.to_native(0)
    net.set_mode(5)
    p, v = net(x.cuda())
    torch.cuda.synchronize()
    idx = np.arange(batch) if batch <= 64 else np.random.RandomState(0).choice(batch, 64, replace=False)
    with torch.no_grad():
        rp, rv = ref(x[idx])
    assert np.abs(p.cpu().numpy()[idx] - rp.numpy()).max() < TOL
    assert np.abs(v.cpu().numpy()[idx] - rv.numpy()).max() < TOL
    net.close()


@pytest.mark.parametrize("mode,rounds", [(5, 1200), (2, 400)])
def test_resident_trunk_stress_alternating_inputs(mode, rounds):
    """The group-resident kernels hand a group's activations from layer to layer through HBM/L2 with only a
    workgroup barrier + workgroup-scope acquire in between (the group is private to one workgroup, whose waves
    share one L1). A stale line would be a timing-dependent error, so: 8 different 4096-board inputs in random
    order, `rounds` forwards back to back under full load; every result must equal the first result of the same
    input bit for bit, and those agree with the per-layer fp32 path (mode 4: one launch per layer, no in-kernel
    hand-off) to fp32 accuracy."""
    import torch
    from alpha_omok_amd.pvnet import PVNet
    nb, B, planes, batch = 4, 9, 128, 4096
    torch.manual_seed(11)
    ref = PVNet(nb, 5, planes, B).eval()
    net = ref.to_native(0)
    rs = np.random.RandomState(5)
    xs = [torch.from_numpy((rs.rand(batch, 5, B, B) < (0.1 + 0.1 * k)).astype(np.float32)).cuda() for k in range(8)]
    net.set_mode(4)
    want = []
    for x in xs:
        p, v = net(x)
        torch.cuda.synchronize()
        want.append((p.clone(), v.clone()))
    net.set_mode(mode)
    first = [None] * len(xs)
    order = rs.randint(0, len(xs), size=rounds)
    bad = 0
    for it, k in enumerate(order):
        p, v = net(xs[k])
        if first[k] is None:
            torch.cuda.synchronize()
            first[k] = (p.clone(), v.clone())
            assert (p - want[k][0]).abs().max().item() < 2e-5 and (v - want[k][1]).abs().max().item() < 2e-5, k
        else:
            if not (torch.equal(p, first[k][0]) and torch.equal(v, first[k][1])):
                bad += 1
    torch.cuda.synchronize()
    assert bad == 0, "%d of %d forwards differed from the first result of the same input" % (bad, rounds)
    net.close()


@pytest.mark.parametrize("batch", [64, 3072])
def test_split_fp16_trunk_reports_activations_beyond_fp16_range(batch):
    """The split-fp16 trunk clamps an activation beyond 65504 (it must stay finite) -- and says so: ao_net_status
    carries AO_NET_FP16_RANGE, and the fused search repeats such a move transparently on the fp32-MFMA trunk (whose
    outputs for the same checkpoint match torch fp32) with the pre-move streams and fresh trees, counting the event
    instead of raising. batch 64: per-layer kernel, 3072: resident."""
    import torch
    from alpha_omok_amd.engine import Engine, EngineError, Net
    from alpha_omok_amd.pvnet import PVNet
    B = 9
    sd = pvnet_weights.make_state_dict(2, 5, 128, B, 4)
    net = Net(2, 5, 128, B, 0)
    net.load_state_dict(sd)
    net.set_mode(5)
    x = torch.from_numpy((np.random.RandomState(1).rand(batch, 5, B, B) < 0.4).astype(np.float32)).cuda()
    net(x)
    assert net.status() == 0                              # an ordinary checkpoint stays in range
    big = dict(sd)
    big["bn1.weight"] = (sd["bn1.weight"] * 3.0e5).astype(np.float32)   # conv1 output ~1e5: beyond fp16
    net.load_state_dict(big)
    p, v = net(x)
    assert np.isfinite(p.cpu().numpy()).all() and np.isfinite(v.cpu().numpy()).all()
    assert net.status(clear=False) == 1 and net.status() == 1 and net.status() == 0
    if batch != 64:
        return
    # the engine repeats such a move on the fp32-MFMA trunk -- pre-move streams restored, fresh trees -- and says so
    net.set_mode(0)
    S = 6
    eng = Engine(B, S, 5, games=batch, noise=True)
    eng.seed_all(np.arange(batch, dtype=np.uint32) + 50)
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        pi, vis, pol = eng.search(net, tau=1)
    assert eng.fp16_range_events() == (1, batch)
    assert np.all(vis.sum(axis=1) == S) and net.status() == 0   # fresh roots: S + 1 simulations, S child visits
    rng_after = [eng.get_rng_state(g) for g in (0, batch - 1)]
    # ... and the result is the one a search on the fp32-MFMA trunk gives from the same streams and positions
    net.set_mode(2)
    eng2 = Engine(B, S, 5, games=batch, noise=True)
    eng2.seed_all(np.arange(batch, dtype=np.uint32) + 50)
    pi2, vis2, pol2 = eng2.search(net, tau=1)
    assert eng2.fp16_range_events() == (0, 0)
    np.testing.assert_array_equal(vis, vis2)
    np.testing.assert_array_equal(pol, pol2)
    np.testing.assert_array_equal(pi, pi2)
    for g, st in zip((0, batch - 1), rng_after):
        st2 = eng2.get_rng_state(g)
        assert np.array_equal(st[0], st2[0]) and st[1:] == st2[1:]
    # (this artificial checkpoint saturates the policy a ply later -- NaN priors, as in the reference -- so the next
    # events are provoked from reset roots.) The network goes back to its mode after an event; from the third event on
    # it stays on the fp32-MFMA trunk.
    from alpha_omok_amd import _lib
    L = _lib.load()
    net.set_mode(0)
    assert L.ao_net_get_mode(net._h) == 0
    for k in (2, 3):
        eng.reset()
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            eng.search(net, tau=1)
        assert eng.fp16_range_events() == (k, k * batch)
        assert L.ao_net_get_mode(net._h) == (0 if k < 3 else 2)
    eng.reset()
    eng.search(net, tau=1)                               # mode 2 now: no event, no warning
    assert eng.fp16_range_events() == (3, 3 * batch)
    ref = PVNet(2, 5, 128, B)
    ref.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in big.items()})
    ref.eval()
    with torch.no_grad():
        rp, rv = ref(x.cpu())
    p2, v2 = net(x)
    assert np.abs(p2.cpu().numpy() - rp.numpy()).max() < TOL and np.abs(v2.cpu().numpy() - rv.numpy()).max() < TOL
    # the fallback belongs to the weights that overflowed: loading other weights returns the network to the mode that was
    # asked for (0 here), and a search with them raises no event
    assert net.get_mode() == 2
    net.load_state_dict(sd)
    assert net.get_mode() == 0
    eng.reset()
    eng.search(net, tau=1)
    assert eng.fp16_range_events() == (3, 3 * batch) and net.get_mode() == 0
    for en in (eng, eng2):
        en.close()


@pytest.mark.parametrize("B,G,S,nb,C", [(9, 64, 24, 2, 5), (9, 3072, 12, 1, 5), (15, 40, 16, 2, 5), (9, 50, 20, 1, 7), (5, 80, 20, 1, 3)])
def test_fused_search_on_bit_planes_equals_stepwise_on_float_planes(B, G, S, nb, C):
    """ao_search hands the split-fp16 kernels the leaf planes as BITS (one byte per cell, written by the tree kernel;
    k_trunk16hb / k_layer16h<.., 2>), the stepwise protocol hands the same network fp32 NCHW planes through
    ao_net_forward (k_nchw_to_il + the fp32-plane kernels). The planes are 0/1, so conv1 sees identical operands and
    the two searches must agree bit for bit -- visits, priors, moves -- on the per-layer kernel (G = 64, 15 x 15),
    the resident kernel (3072 boards = 192 groups) and other history depths (C = 3, 7)."""
    import torch
    from alpha_omok_amd.engine import Engine
    from gpu_helpers import HostEvalRunner
    net = _native(nb, C, 128, B, 9)
    net.set_mode(5)
    seeds = np.arange(40, 40 + G, dtype=np.uint32)
    e1 = Engine(B, S, C, games=G, noise=True)
    e2 = Engine(B, S, C, games=G, noise=True)
    e1.seed_all(seeds)
    e2.seed_all(seeds)
    run = HostEvalRunner(e2)
    for t in range(3):
        tau = np.full(G, 1 if t < 2 else 0, np.int8)
        pi1, vis1, pol1 = e1.search(net, tau=tau)
        e2.begin_move()
        while e2.sims_left() > 0:
            e2.collect_leaves(run.planes.data_ptr())
            e2.sync()
            p, v = net(run.planes)
            torch.cuda.synchronize()
            e2.apply_evals(p.data_ptr(), v.data_ptr())
        pi2, vis2, pol2 = e2.end_move(tau)
        np.testing.assert_array_equal(vis1, vis2)
        np.testing.assert_array_equal(pol1, pol2)
        np.testing.assert_array_equal(pi1, pi2)
        a1, w1 = e1.play()
        a2, w2 = e2.play()
        np.testing.assert_array_equal(a1, a2)
        np.testing.assert_array_equal(w1, w2)
    assert net.status() == 0
    e1.close()
    e2.close()
    net.close()


@pytest.mark.parametrize("mode,planes", [(4, 64), (2, 64), (3, 64), (5, 128)])
def test_fused_search_packs_active_games_only(mode, planes):
    """ao_search runs the network on the ACTIVE games only (rows packed to the front of the batch). With a kernel family
    whose per-board result does not depend on the batch (modes 2 / 3 / 4; mode 5 with every move on the per-layer
    kernel) the packed search must equal the stepwise protocol that evaluates all G slots, bit for bit, under
    changing active masks -- and inactive games must stay untouched."""
    import torch
    from alpha_omok_amd.engine import Engine
    from gpu_helpers import HostEvalRunner
    B, S, G = 9, 20, 70
    net = _native(2, 5, planes, B, 21)
    net.set_mode(mode)
    seeds = np.arange(900, 900 + G, dtype=np.uint32)
    e1 = Engine(B, S, 5, games=G, noise=True)
    e2 = Engine(B, S, 5, games=G, noise=True)
    e1.seed_all(seeds)
    e2.seed_all(seeds)
    run = HostEvalRunner(e2)
    rs = np.random.RandomState(3)
    for t in range(4):
        active = np.ones(G, np.uint8) if t == 0 else (rs.rand(G) < (0.6, 0.25, 0.05)[t - 1]).astype(np.uint8)
        if t == 3:
            active[:] = 0
            active[[7, 41]] = 1
        tau = np.ones(G, np.int8)
        before = [e1.get_moves(g) for g in range(G)]
        pi1, vis1, pol1 = e1.search(net, tau=tau, active=active)
        e2.begin_move(active)
        while e2.sims_left() > 0:
            e2.collect_leaves(run.planes.data_ptr())
            e2.sync()
            p, v = net(run.planes)
            torch.cuda.synchronize()
            e2.apply_evals(p.data_ptr(), v.data_ptr())
        pi2, vis2, pol2 = e2.end_move(tau)
        on = active.astype(bool)
        np.testing.assert_array_equal(vis1[on], vis2[on])
        np.testing.assert_array_equal(pol1[on], pol2[on])
        np.testing.assert_array_equal(pi1[on], pi2[on])
        a1, w1 = e1.play()
        a2, w2 = e2.play()
        np.testing.assert_array_equal(a1, a2)
        assert (a1[~on] == -1).all()
        for g in range(G):
            assert len(e1.get_moves(g)) == len(before[g]) + int(active[g])
    e1.close()
    e2.close()
    net.close()


@pytest.mark.parametrize("B,G,S,nb,C", [(9, 1, 64, 2, 5), (9, 5, 48, 2, 5), (15, 3, 24, 1, 5), (9, 40, 16, 1, 7), (5, 7, 20, 1, 3)])
def test_fused_per_game_step_equals_separate_launches(B, G, S, nb, C, monkeypatch):
    """A few games on a 128-plane network: ao_search runs heads + expansion / backup / selection + the next leaf's conv1 as ONE
    launch per game (k_step_board, step_kernels.hip). The tree code is the same device code; conv1 is formulated differently
    (K = tap * 8 + plane on fp16 MFMAs with split weights instead of fp32 MFMAs per tap), exact products of 0/1 planes with
    fp32 accumulation either way. The searches must agree with the three-launch form (AO_FUSED_STEP=0): same visits, priors,
    moves and stream positions over several plies, with a changing active mask."""
    from alpha_omok_amd.engine import Engine
    net = _native(nb, C, 128, B, 31)
    seeds = np.arange(70, 70 + G, dtype=np.uint32)
    e1 = Engine(B, S, C, games=G, noise=True)
    e2 = Engine(B, S, C, games=G, noise=True)
    e1.seed_all(seeds)
    e2.seed_all(seeds)
    rs = np.random.RandomState(1)
    for t in range(4):
        tau = np.full(G, 1 if t < 2 else 0, np.int8)
        active = np.ones(G, np.uint8)
        if t == 2 and G > 2:
            active = (rs.rand(G) < 0.6).astype(np.uint8)
            active[0] = 1
        monkeypatch.delenv("AO_FUSED_STEP", raising=False)
        pi1, vis1, pol1 = e1.search(net, tau=tau, active=active)
        monkeypatch.setenv("AO_FUSED_STEP", "0")
        pi2, vis2, pol2 = e2.search(net, tau=tau, active=active)
        on = active.astype(bool)
        np.testing.assert_array_equal(vis1[on], vis2[on])
        np.testing.assert_allclose(pol1[on], pol2[on], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(pi1[on], pi2[on])
        a1, w1 = e1.play()
        a2, w2 = e2.play()
        np.testing.assert_array_equal(a1, a2)
        np.testing.assert_array_equal(w1, w2)
        for g in range(G):
            assert e1.get_rng_state(g)[1] == e2.get_rng_state(g)[1]
    assert net.status() == 0
    e1.close()
    e2.close()
    net.close()


@pytest.mark.parametrize("nb,B,batch", [(2, 15, 130), (10, 15, 300), (3, 11, 129), (2, 13, 200), (1, 10, 128), (2, 15, 70)])
def test_board_resident_trunk_wide_boards_vs_torch_and_per_layer(nb, B, batch):
    """k_boardh<B> (round 5, net_board_h16.hpp): boards wider than 9, from 64 boards on -- a workgroup carries ONE board through
    all trunk convs, activations resident in LDS, cells as the MFMA N dimension, column shifts as DPP row shifts. Against the
    torch fp32 module (model.py:76-104) within the 1e-4 of BASELINE.json's north star, against the per-layer kernels
    (AO_BOARDK=0: same split-fp16 arithmetic, another summation order) within 2e-5, and with a batch that is not a multiple of
    16 or of the workgroup count (boards loop, partial last group). The dominant kernel is named by the library's own planning."""
    import os
    import torch
    from alpha_omok_amd.engine import plan_kernel
    from alpha_omok_amd.pvnet import PVNet
    assert plan_kernel(nb, 5, 128, B, batch, in_kind=1)[0].startswith("k_boardh<%d," % B)
    assert plan_kernel(nb, 5, 128, B, 63, in_kind=1)[0].startswith("k_layer16h<%d>" % B)
    assert plan_kernel(nb, 5, 128, B, batch, in_kind=1, trunk_mode=6)[0].startswith("k_layer16h<%d>" % B)   # one arithmetic for every batch size
    torch.manual_seed(nb * 100 + B)
    ref = PVNet(nb, 5, 128, B)       # PyTorch default init (the deterministic generator saturates a 10-block stack)
    with torch.no_grad():
        for m in ref.modules():      # non-trivial BatchNorm statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    ref.eval()
    rs = np.random.RandomState(batch)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    x[:, 4] = (rs.rand(batch, 1, 1) < 0.5).astype(np.float32)
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x))
    net = ref.to_native(0)
    p, v = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert net.dominant_kernel(batch)[0].startswith("k_boardh<%d," % B)
    assert net.status() == 0
    os.environ["AO_BOARDK"] = "0"
    try:
        net_l = ref.to_native(0)
    finally:
        del os.environ["AO_BOARDK"]
    assert net_l.dominant_kernel(batch)[0].startswith("k_layer16h<%d>" % B)
    pl, vl = net_l(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    dp = np.abs(p.cpu().numpy() - rp.numpy()).max()
    dv = np.abs(v.cpu().numpy() - rv.numpy()).max()
    dpl = np.abs(p.cpu().numpy() - pl.cpu().numpy()).max()
    dvl = np.abs(v.cpu().numpy() - vl.cpu().numpy()).max()
    assert dp < TOL and dv < TOL, (dp, dv)
    assert dpl < 2e-5 and dvl < 2e-5, (dpl, dvl)
    # a second forward on the same buffers gives the same bits (the kernel works in place on the conv1 output)
    p2, v2 = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert torch.equal(p, p2) and torch.equal(v, v2)
    net.close()
    net_l.close()


@pytest.mark.parametrize("nb,B,planes,batch", [(2, 9, 100, 300), (1, 7, 40, 33), (2, 9, 200, 64), (2, 15, 100, 70), (1, 9, 300, 48), (1, 5, 500, 9)])
def test_widths_that_are_not_a_multiple_of_32_run_natively_zero_padded(nb, B, planes, batch):
    """model.PVNet takes any `planes` (model.py:76-85). Round 5: widths up to 256 that are not a multiple of 32 are exported
    zero-padded to the next multiple (pvnet.pad_state_dict) and run on the same MFMA kernels -- 100 planes on the split-fp16
    kernels at 128, 200 on the fp32-MFMA layer kernels at 224 -- instead of falling back to the torch module; through
    PVNet.to_native and through the engine's Evaluator (no warning, a native Net). Against torch fp32 within 1e-4."""
    import warnings
    import torch
    from alpha_omok_amd.evaluator import Evaluator
    from alpha_omok_amd.pvnet import PVNet, native_width
    torch.manual_seed(planes + nb)
    ref = PVNet(nb, 5, planes, B)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    ref.eval()
    rs = np.random.RandomState(batch)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    with torch.no_grad():
        rp, rv = ref(torch.from_numpy(x))
    net = ref.to_native(0)
    assert net.planes == native_width(planes) and net.planes % 32 == 0
    p, v = net(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    dp, dv = np.abs(p.cpu().numpy() - rp.numpy()).max(), np.abs(v.cpu().numpy() - rv.numpy()).max()
    assert dp < TOL and dv < TOL, (dp, dv)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ev = Evaluator(0)
        n2 = ev.native_net(ref.cuda(), B, 5)
    assert n2 is not None and n2.planes == native_width(planes)
    p2, v2 = n2(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert torch.equal(p2, p) and torch.equal(v2, v)
    net.close()
