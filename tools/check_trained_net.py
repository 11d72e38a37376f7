"""Evidence on a TRAINED network (a checkpoint written by tools/train_omok.py), taken on the GPU box:

  (1) forward error of every kernel family on real self-play positions: 4096 boards sampled from games the network
      plays against itself -- resident split-fp16 trunk with 4-byte (fmt 0) and 3-byte (fmt 1) activations, the per-layer
      split-fp16 kernels, the per-board path, the fp32-MFMA trunk -- max |dp| and max |dv| against torch fp32 (the
      tolerance north_star states is 1e-4 against the fp32 network) and against a float64 evaluation of the same weights;
  (2) what a sharp policy does to the engine over one main.self_play(GAMES): fp16-range events (ao_fp16_range_events),
      arena trims at the default node_cap (ao_trim_stats), mean selection depth, terminal-leaf share, tree nodes kept.

    python tools/check_trained_net.py --ckpt profiles/r4_trained_9x9_4block.pt --blocks 4 --out gpurun_out/r4_trained_net.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--planes", type=int, default=128)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--boards", type=int, default=4096, help="positions in the forward comparison")
    ap.add_argument("--games", type=int, default=4096, help="games of the self-play pass (0: skip)")
    ap.add_argument("--position-games", type=int, default=512)
    ap.add_argument("--position-sims", type=int, default=100)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    import torch
    import alpha_omok_amd.main as m
    from alpha_omok_amd.pvnet import PVNet

    B = a.board
    ref = PVNet(a.blocks, 5, a.planes, B)
    ref.load_state_dict(torch.load(a.ckpt, map_location="cpu"))
    ref.eval()
    out = dict(ckpt=os.path.basename(a.ckpt), blocks=a.blocks, planes=a.planes, board=B, device=torch.cuda.get_device_name(0))

    # ---- real positions: the network's own self-play games (tau schedule and noise as in training)
    m.configure(board_size=B, n_mcts=a.position_sims, n_blocks=a.blocks, out_planes=a.planes, seed=77,
                model=PVNet(a.blocks, 5, a.planes, B).cuda())
    m.Agent.model.load_state_dict(ref.state_dict())
    m.cur_memory.clear()
    m.self_play(a.position_games)
    states = np.stack([s for s, _, _ in m.cur_memory]).astype(np.float32)
    m.cur_memory.clear()
    m.release_engine()
    rs = np.random.RandomState(5)
    pick = rs.choice(states.shape[0], min(a.boards, states.shape[0]), replace=False)
    x = torch.from_numpy(states[pick])
    stones = x[:, :4].sum(dim=(1, 2, 3))
    out["positions"] = dict(n=int(x.shape[0]), from_games=a.position_games, from_samples=int(states.shape[0]),
                            mean_stones_on_latest_planes=float(stones.mean()))
    with torch.no_grad():
        p32, v32 = ref(x)                                  # torch fp32 (CPU): the reference arithmetic
        sub = rs.choice(x.shape[0], min(256, x.shape[0]), replace=False)
        p64, v64 = ref.double()(x[sub].double())
        ref.float()
    out["policy_sharpness"] = dict(mean_max_prior=float(p32.max(dim=1).values.mean()), mean_abs_value=float(v32.abs().mean()),
                                   max_abs_value=float(v32.abs().max()))

    def run(mode, fmt=None, boards=None):
        if fmt is not None:
            os.environ["AO_TRUNK_FMT"] = str(fmt)
        net = ref.to_native(0)
        os.environ.pop("AO_TRUNK_FMT", None)
        net.set_mode(mode)
        xb = x if boards is None else x[:boards]
        p, v = net(xb.cuda())
        torch.cuda.synchronize()
        name = net.dominant_kernel(xb.shape[0])[0].split(" (")[0]
        st = net.status()
        net.close()
        p, v = p.cpu(), v.cpu()
        n = xb.shape[0]
        in64 = sub[sub < n]
        where = {int(s): i for i, s in enumerate(sub)}
        rows64 = [where[int(s)] for s in in64]
        r = dict(kernel=name, boards=int(n), fp16_range_flag=int(st),
                 max_dp_vs_torch_fp32=float((p - p32[:n]).abs().max()), max_dv_vs_torch_fp32=float((v - v32[:n]).abs().max()))
        if len(in64):
            r["max_dp_vs_fp64"] = float((p[in64].double() - p64[rows64]).abs().max())
            r["max_dv_vs_fp64"] = float((v[in64].double() - v64[rows64]).abs().max())
        return r

    fw = {}
    fw["resident_fmt0_4byte"] = run(5, fmt=0)
    fw["resident_fmt1_3byte"] = run(5, fmt=1)
    fw["per_layer_split_fp16"] = run(6)
    fw["per_layer_split_fp16_1024_boards"] = run(6, boards=1024)
    fw["ksplit_1024_boards"] = run(0, boards=1024)          # mode 0 = the planner: k_layer16hk at 48 .. 64 groups
    fw["row_kernel_512_boards"] = run(0, boards=512)        # ... k_row16hk below (33 boards .. 47 groups)
    fw["row_kernel_100_boards"] = run(0, boards=100)
    fw["per_board_64_boards"] = run(3, boards=64)
    fw["fp32_mfma_trunk"] = run(2)
    with torch.no_grad():
        fw["torch_fp32_vs_fp64"] = dict(max_dp=float((p32[sub].double() - p64).abs().max()), max_dv=float((v32[sub].double() - v64).abs().max()))
    out["forward"] = fw
    for k, r in fw.items():
        print(k, json.dumps(r), flush=True)

    # ---- one self-play pass at full size with the trained network: what the sharp policy does to the trees
    if a.games > 0:
        m.configure(board_size=B, n_mcts=a.sims, n_blocks=a.blocks, out_planes=a.planes, seed=123,
                    model=PVNet(a.blocks, 5, a.planes, B).cuda())
        m.Agent.model.load_state_dict(ref.state_dict())
        m.cur_memory.clear()
        t0 = time.time()
        r = m.self_play(a.games)
        dt = time.time() - t0
        eng = m._engine
        st = m.search_totals
        sims = max(st["evaluated"] + st["terminal"], 1)
        ev, evg = eng.fp16_range_events()
        out["self_play"] = dict(games=a.games, sims=a.sims, moves=r["moves"], seconds=round(dt, 2), moves_per_s=round(r["moves"] / dt, 1),
                                mean_game_len=round(r["moves"] / max(r["episodes"], 1), 2), result=dict(m.result),
                                mean_select_depth=round(st["levels"] / sims, 3), terminal_leaf_share=round(st["terminal"] / sims, 4),
                                node_cap=eng.node_cap()[0], trims=dict(m.trim_stats), fp16_range_events=ev, fp16_games_redone=evg,
                                net_mode_after=m._evaluator._net.get_mode() if m._evaluator._net is not None else None)
        print("self_play", json.dumps(out["self_play"]), flush=True)
        m.release_engine()
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
