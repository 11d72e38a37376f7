"""-m gpu: the TWO-product split-fp16 conv kernels (net_w16.hip; ao_net_products).

The split-fp16 kernels compute the 3x3 convs of model.PVNet (model.py:6-31,97-104) as x*w = xh*wh + xh*wl + xl*wh on fp16 halves,
fp32 accumulate. When every conv weight (times its layer's power of two) IS an fp16 number, wl == 0 and the middle product adds
exact zeros: the library notices at ao_net_finalize and plans kernels that leave it out. What must hold:
  * which networks qualify is decided by the weights alone (arbitrary fp32 weights: three products, as ever);
  * on qualifying weights the two-product kernels return the SAME BITS as the three-product kernels, in every kernel family that
    has a two-product form (resident trunk on float planes and on the engine's bit planes, the three per-layer kernels, the
    board-resident trunk);
  * and they are the fp32-equivalent contraction: within 1e-5 of torch fp32 on the same weights (the repo-wide contract is 1e-4);
  * training with configure(fp16_grid_weights=True) keeps the conv weights there, so the searches of a training run stay on two
    products, and the checkpoint is a plain fp32 state_dict.
"""
import numpy as np
import pytest

import pvnet_weights

pytestmark = pytest.mark.gpu


def _grid_sd(nb, planes, B, seed, project=True):
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, seed)
    if project:
        for k, v in sd.items():
            if v.ndim == 4 and v.shape[2] == 3:
                sd[k] = v.astype(np.float16).astype(np.float32)
    return sd


def _planes(batch, B, seed):
    rs = np.random.RandomState(seed)
    x = (rs.rand(batch, 5, B, B) < 0.3).astype(np.float32)
    x[:, 4] = (rs.rand(batch, 1, 1) < 0.5).astype(np.float32)
    return x


# (blocks, board, batch, mode, the kernel the library must plan)
CASES = [
    (2, 9, 3200, 0, "k_trunk16h_w16<9, 4, 0>"),      # >= 192 groups: the resident trunk (float planes: ao_net_forward)
    (1, 7, 3104, 0, "k_trunk16h_w16<7, 4, 0>"),
    (2, 9, 1500, 6, "k_layer16h_w16<9>"),            # one launch per conv (mode 6: the per-layer family for every batch size)
    (2, 9, 1500, 0, "k_layer16h_w16<9>"),            # ... and as planned for a medium batch
    (1, 5, 400, 6, "k_layer16h_w16<5>"),
    (2, 9, 900, 0, "k_layer16hk_w16<9, 4>"),         # 48 .. 64 groups: a group split over four workgroups by cout pairs
    (1, 7, 1024, 0, "k_layer16hk_w16<7, 4>"),
    (2, 9, 300, 0, "k_row16hk_w16<9>"),              # below 48 groups: one workgroup per group x output row x cout pair
    (1, 6, 500, 0, "k_row16hk_w16<6>"),
    (2, 9, 16, 0, "k_conv_cells_h_w16<9, 8>"),       # a handful of boards: the per-board path (half the weight bytes through the CU)
    (1, 15, 8, 0, "k_conv_cells_h_w16<15, 8>"),
    (2, 15, 96, 0, "k_boardh_w16<15, 1>"),           # wide boards: one board per workgroup, resident in LDS
    (1, 11, 130, 0, "k_boardh_w16<11, 1>"),
    (2, 15, 40, 0, "k_layer16h_w16<15>"),            # few wide boards: the per-layer kernel on column tiles
    (1, 13, 33, 6, "k_layer16h_w16<13>"),
]


@pytest.mark.parametrize("nb,B,batch,mode,kernel", CASES)
def test_two_products_are_the_three_product_bits_and_the_fp32_contraction(nb, B, batch, mode, kernel):
    import torch
    from alpha_omok_amd.engine import Net
    from alpha_omok_amd.pvnet import PVNet
    sd = _grid_sd(nb, 128, B, 300 + nb + B)
    net = Net(nb, 5, 128, B, 0)
    net.load_state_dict(sd)
    net.set_mode(mode)
    assert net.products() == (2, True)
    x = torch.from_numpy(_planes(batch, B, batch)).cuda()
    p2, v2 = net(x)
    torch.cuda.synchronize()
    assert net.dominant_kernel(batch)[0].startswith(kernel), net.dominant_kernel(batch)[0]
    assert "2 products" in net.dominant_kernel(batch)[0]
    assert net.products(3) == (3, True)
    p3, v3 = net(x)
    torch.cuda.synchronize()
    assert "_w16" not in net.dominant_kernel(batch)[0] and "3 products" in net.dominant_kernel(batch)[0]
    assert torch.equal(p2, p3) and torch.equal(v2, v3), "the two-product kernel differs from the three-product kernel on fp16 weights"
    assert net.products(0) == (2, True)
    assert net.status() == 0
    ref = PVNet(nb, 5, 128, B)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref.eval()
    with torch.no_grad():
        rp, rv = ref(x.cpu())
    dp = float((p2.cpu() - rp).abs().max())
    dv = float((v2.cpu() - rv).abs().max())
    assert dp < 1e-5 and dv < 1e-5, (dp, dv)
    net.close()


def test_arbitrary_fp32_weights_stay_on_three_products():
    """The library decides from the weights: one conv weight off the fp16 grid -- in conv1 or in any ResBlock -- and the network
    runs on three products; asking for two is not possible (request 0 = automatic, 3 = force three)."""
    from alpha_omok_amd.engine import EngineError, Net
    net = Net(2, 5, 128, 9, 0)
    net.load_state_dict(_grid_sd(2, 128, 9, 5, project=False))
    assert net.products() == (3, False) and net.products(0) == (3, False)
    assert "_w16" not in net.dominant_kernel(4096)[0]
    for key in ("conv1.weight", "layers.1.conv2.weight"):
        sd = _grid_sd(2, 128, 9, 5)
        net.load_state_dict(sd)
        assert net.products() == (2, True)
        w = sd[key].copy()
        w.flat[17] = np.float32(w.flat[17]) * np.float32(1.0 + 2.0 ** -13)      # one weight, 2^-13 off its fp16 value
        assert np.float32(np.float16(w.flat[17])) != w.flat[17]
        sd[key] = w
        net.load_state_dict(sd)
        assert net.products() == (3, False), key
    with pytest.raises(EngineError):
        net.products(2)
    net.close()


@pytest.mark.parametrize("G,kernel2,kernel3", [(3072, "k_trunk16hb_w16<9, 4, 0>", "k_trunk16hb<9, 4, 0>"),
                                               (6, "k_conv_cells_h_w16<9, 8>", "k_conv_cells_h<9, 8>")])
def test_search_on_bit_planes_is_the_same_with_two_and_three_products(G, kernel2, kernel3):
    """ao_search feeds the trunk the engine's bit planes (k_trunk16hb_w16 at 3072+ games; conv1 then runs ONE product: 0/1 planes
    have no low half, fp16 weights have none either) -- or, for a handful of games, runs the fused per-game step around the per-board
    convs: visits, priors and moves of every game equal the three-product search."""
    from alpha_omok_amd.engine import Engine, Net
    B, S = 9, 12
    net = Net(2, 5, 128, B, 0)
    net.load_state_dict(_grid_sd(2, 128, B, 77))
    seeds = np.arange(4000, 4000 + G, dtype=np.uint32)
    out = {}
    for products in (0, 3):
        net.products(products)
        eng = Engine(B, S, 5, games=G, noise=True)
        eng.seed_all(seeds)
        rec = []
        for ply in range(2):
            pi, vis, pol = eng.search(net, tau=np.ones(G, np.int8))
            act, win = eng.play()
            rec.append((pi, vis, pol, act, win))
        out[products] = rec
        kname = net.dominant_kernel(G)[0]
        assert kname.startswith(kernel2 if products == 0 else kernel3), kname
        eng.close()
    for a, b in zip(out[0], out[3]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert np.all(out[0][0][1].sum(axis=1) == S)
    net.close()


def test_fp16_grid_training_keeps_the_conv_weights_on_the_grid():
    """configure(fp16_grid_weights=True) (tools/train_omok.py --fp16-grid-weights): Adam moves fp32 master copies, the module holds
    their fp16 rounding -- after a training pass every 3x3 conv weight is an fp16 number, the masters have moved away from them,
    everything else (BatchNorm, heads) is ordinary fp32, the native export runs on two products, and the checkpoint written in the
    reference's wire format (main.py:339-365) loads back to the same two-product network."""
    import torch
    import alpha_omok_amd.main as main
    B = 9
    try:
        torch.manual_seed(3)
        main.configure(board_size=B, n_mcts=8, n_blocks=2, in_planes=5, out_planes=128, seed=3, fp16_grid_weights=True,
                       carry_over=False, oversubscribe=1.0, rows='auto', overlap_train=False, device_replay=False, strict=False)
        main.TRAIN_STEPS, main.BATCH_SIZE = 6, 32
        main.rep_memory.clear()
        main.cur_memory.clear()
        model = main.Agent.model
        convs = {n: p for n, p in model.named_parameters() if p.dim() == 4 and p.shape[2] == 3}
        assert len(convs) == 5 and len(main._grid_masters) == 5
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        for n, p in convs.items():
            assert torch.equal(p.detach().half().float(), p.detach()), n       # on the grid from the start
        main.self_play(12)
        net = main._evaluator._net
        assert net.products() == (2, True)
        losses = main.train(1, 1)
        assert len(losses) == 6 and all(np.isfinite(l).all() for l in losses) and main.skipped_steps == 0
        moved = 0
        for n, p in model.named_parameters():
            if n in convs:
                assert torch.equal(p.detach().half().float(), p.detach()), n
            moved += int(not torch.equal(p.detach(), before[n]))
        assert moved >= len(before) - 2                                        # the pass trained (nearly) every tensor
        off_grid = sum(int(not torch.equal(m.half().float(), m)) for _, m in main._grid_masters)
        assert off_grid == 5                                                   # the masters are ordinary fp32 numbers
        main.self_play(4)                                                      # re-export after the pass: still two products
        assert main._evaluator._net.products() == (2, True)
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            path = main.save_model(main.Agent, 3, main.step, directory=d, datetime_now="180927")
            sd = torch.load(path, map_location="cpu", weights_only=True)
            assert all(v.dtype == torch.float32 for k, v in sd.items() if v.is_floating_point())
            from alpha_omok_amd.pvnet import PVNet
            fresh = PVNet(2, 5, 128, B)
            fresh.load_state_dict(sd)
            nat = fresh.to_native(0)
            assert nat.products() == (2, True)
            nat.close()
    finally:
        main.TRAIN_STEPS = None
        main.configure(board_size=B, n_mcts=8, n_blocks=2, out_planes=128, seed=0, fp16_grid_weights=False)
        main.release_engine()
    assert main._grid_masters == []


def test_a_non_finite_gradient_contributes_nothing_and_is_counted():
    """main.train_batch (main.py:283-305): -(pi * p.log()) is -inf * pi once the softmax underflows to an exact 0 on a visited move;
    the reference would write NaN into every weight. Here the gradient of such a mini-batch is replaced by zeros on the device (no
    host synchronisation per step): no weight becomes NaN, the step is counted in main.skipped_steps once the pass is over."""
    import torch
    import alpha_omok_amd.main as main
    B = 5
    try:
        torch.manual_seed(1)
        main.configure(board_size=B, n_mcts=4, n_blocks=1, in_planes=5, out_planes=32, seed=1, carry_over=False, oversubscribe=1.0,
                       overlap_train=False, device_replay=False, strict=False, fp16_grid_weights=False)
        main.TRAIN_STEPS, main.BATCH_SIZE = 3, 8
        main.rep_memory.clear()
        main.cur_memory.clear()
        main.self_play(6)
        with torch.no_grad():                                                  # a policy head that underflows to exact zeros
            main.Agent.model.policy_head.policy_fc.bias[0] = 1e4
        s0 = main.skipped_steps
        losses = main.train(1, 1)
        assert main.skipped_steps == s0 + 3 and len(losses) == 3
        assert all(torch.isfinite(p).all() for p in main.Agent.model.parameters())
    finally:
        main.TRAIN_STEPS = None
        main.BATCH_SIZE = 32
        main.configure(board_size=9, n_mcts=8, n_blocks=2, out_planes=128, seed=0)
        main.release_engine()
