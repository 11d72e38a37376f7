"""Train a PVNet with the engine itself -- the reference's main.py loop (main.py:377-414: self-play -> train -> repeat)
at the scale the MI355X engine is built for -- and record the evidence that it learns.

One iteration = main.self_play(GAMES) (thousands of concurrent games, device replay) + main.train() with
main.TRAIN_STEPS mini-batches of main.BATCH_SIZE; every --eval-every iterations the current network plays
--eval-matches head-to-head games against the iteration-0 network through evaluate.evaluate_batched (eval_main.py:191-198
ELO, K = 32, both sides start at 1500; noise off, tau 0). Runs ON THE GPU BOX:

    python tools/train_omok.py --out gpurun_out/r4_train --minutes 50 --board 9 --blocks 4 --sims 400 --games 2048

Writes <out>/log.jsonl (one line per iteration: games, move decisions/s, mean game length, results, loss, mean selection
depth, terminal-leaf share, arena trims, fp16-range events, and the ELO lines), <out>/ckpt_<iter>.pt (reference wire
format: torch.save(state_dict)) for the listed iterations and <out>/final.pt.
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
    ap.add_argument("--out", default="gpurun_out/train")
    ap.add_argument("--minutes", type=float, default=10.0, help="wall-clock budget of the self-play/train loop")
    ap.add_argument("--iters", type=int, default=10 ** 9)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--planes", type=int, default=128)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--games", type=int, default=2048, help="self-play games per iteration")
    ap.add_argument("--first-games", type=int, default=None, help="games of iteration 0 (default: --games)")
    ap.add_argument("--steps", type=int, default=300, help="mini-batches per iteration (main.TRAIN_STEPS)")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--l2", type=float, default=1e-4)
    ap.add_argument("--memory", type=int, default=2_000_000, help="replay entries (8 per sample)")
    ap.add_argument("--no-carry-over", action="store_true",
                    help="play every iteration's games to the end before training (the reference's strict alternation, main.py:250-262). "
                         "Default: carry-over -- the engine stays full across iterations, a slot whose game ends starts an episode of the "
                         "NEXT iteration(s), so an episode may be played partly under the weights of up to two iterations ago (the usual "
                         "asynchronous-actor trade; alpha_omok_amd.main.configure)")
    ap.add_argument("--oversubscribe", type=float, default=1.25,
                    help="game slots per row of the 4096-row evaluation batch (main.configure: 1.25 = 5120 resident games; terminal leaves "
                         "take no row, the extra games fill what they leave). 1 = as many games as rows")
    ap.add_argument("--rows", default="auto", choices=("auto", "static", "dynamic"))
    ap.add_argument("--rows-per-sim", type=int, default=None,
                    help="rows of the evaluation batch = leaves evaluated per simulation (main.MAX_CONCURRENT; default 4096). BASELINE configs[4]'s "
                         "per-GPU shape is 1024 (15x15: 256 workgroups of k_boardh x 4 boards)")
    ap.add_argument("--eval-every", type=int, default=5)
    ap.add_argument("--eval-matches", type=int, default=64)
    ap.add_argument("--eval-sims", type=int, default=None)
    ap.add_argument("--yardstick", default=None, help="'puct:400' / 'uct:400': also play every evaluation against the rollout agent "
                                                      "(agents.py:263-634) with that many simulations -- a fixed opponent that needs no network")
    ap.add_argument("--eval-dense-until", type=int, default=0, help="evaluate after EVERY iteration up to this one (the steep part of the curve)")
    ap.add_argument("--ckpt-every", type=int, default=10)
    ap.add_argument("--max-ckpts", type=int, default=6, help="checkpoints kept on disk besides iteration 0 and final (gpurun_out is merged back up to 64 MiB)")
    ap.add_argument("--host-states", action="store_true", help="build the samples' state planes on the host and upload them (main.DEVICE_STATES = False): the path before device-side sample emission")
    ap.add_argument("--overlap-train", action="store_true", default=True,
                    help="(default since the end of round 5) main.train_async: an iteration's training pass runs on a worker thread and a side stream "
                         "while the NEXT iteration's games are played with the weights exported before it started (one more iteration of staleness "
                         "than carry-over already has; profiles/r5zb_overlap_train.txt, r5zz_overlap_learning_check.txt)")
    ap.add_argument("--no-overlap-train", dest="overlap_train", action="store_false",
                    help="the reference's alternation: the games wait for main.train (main.py:377-414)")
    ap.add_argument("--fp16-grid-weights", action="store_true",
                    help="keep the 3x3 conv weights on the fp16 grid (main.configure(fp16_grid_weights=True): fp32 master copies in Adam, the module holds "
                         "their fp16 rounding) -- such a network runs on the two-product split-fp16 kernels (ao_net_products); the checkpoint stays a "
                         "plain fp32 state_dict the reference loads")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--resume", default=None, help="state_dict to start from")
    a = ap.parse_args()

    import torch
    import alpha_omok_amd.main as m
    from alpha_omok_amd import evaluate
    from alpha_omok_amd.pvnet import PVNet

    os.makedirs(a.out, exist_ok=True)
    log = open(os.path.join(a.out, "log.jsonl"), "a")

    def emit(rec):
        log.write(json.dumps(rec) + "\n")
        log.flush()
        print(json.dumps(rec), flush=True)

    m.BATCH_SIZE, m.LR, m.L2, m.MEMORY_SIZE, m.TRAIN_STEPS = a.batch, a.lr, a.l2, a.memory, a.steps
    m.DEVICE_STATES = not a.host_states
    if a.rows_per_sim:
        m.MAX_CONCURRENT = a.rows_per_sim
    m.configure(board_size=a.board, n_mcts=a.sims, n_blocks=a.blocks, out_planes=a.planes, seed=a.seed,
                device_replay=True, carry_over=not a.no_carry_over, oversubscribe=a.oversubscribe, rows=a.rows,
                fp16_grid_weights=a.fp16_grid_weights)
    if a.resume:
        m.Agent.model.load_state_dict(torch.load(a.resume, map_location=m.device))
        m._grid_sync()                                   # (fp16-grid mode: masters = the loaded weights, module = their rounding)
    dev = m.device
    base = PVNet(a.blocks, m.IN_PLANES, a.planes, a.board).to(dev)
    base.load_state_dict(m.Agent.model.state_dict())
    base.eval()
    torch.save(base.state_dict(), os.path.join(a.out, "ckpt_0.pt"))
    emit(dict(kind="config", **vars(a), device=torch.cuda.get_device_name(0)))

    if a.yardstick:                                       # where the untrained network stands against the fixed opponent
        kind, ysims = a.yardstick.split(":")
        t0 = time.time()
        yr, (ype, yee), yg = evaluate.evaluate_batched(base, kind, a.board, a.eval_sims or a.sims, n_mcts_enemy=int(ysims),
                                                       n_match=a.eval_matches, seed=9000, device=0)
        emit(dict(kind="elo", iter=0, vs=a.yardstick, matches=a.eval_matches, result=yr, elo_gain_reference_K32=round(ype - 1500.0, 1),
                  score=round((yr["Player"] + 0.5 * yr["Draw"]) / max(sum(yr.values()), 1), 4),
                  mean_plies=round(float(np.mean([len(g[1]) for g in yg])), 1), eval_s=round(time.time() - t0, 2)))
    kept = []
    t_end = time.time() + 60.0 * a.minutes
    games_total = moves_total = 0
    prev = dict(m.search_totals)
    prev_phase = dict(m.phase_seconds)
    prev_model, prev_iter, chain = None, 0, 0.0
    it = 0
    while it < a.iters and time.time() < t_end:
        n = a.games if (it > 0 or a.first_games is None) else a.first_games
        t0 = time.time()
        out = m.self_play(n)
        t_sp = time.time() - t0
        eng = m._engine
        d = {k: m.search_totals[k] - prev[k] for k in prev}   # this call's searches (main.search_totals is cumulative)
        prev = dict(m.search_totals)
        ph = {k: round(m.phase_seconds[k] - prev_phase[k], 3) for k in prev_phase}   # searches / building + appending the samples
        prev_phase = dict(m.phase_seconds)
        lengths = out["moves"] / max(out["episodes"], 1)
        res = dict(m.result)
        t0 = time.time()
        try:
            if a.overlap_train:
                # the PREVIOUS iteration's pass was joined inside self_play (phase 'train_wait'); this one runs beside the next call
                # (so the `loss` of a line is the loss of the pass that ENDED during this iteration: one iteration behind)
                losses = m.last_train_losses or []
                m.last_train_losses = None
                m.train_async(m.N_EPOCHS, it)
                last_iter = it + 1 >= a.iters or time.time() >= t_end
                evaluating = (it + 1) % a.eval_every == 0 or it + 1 <= a.eval_dense_until or (it + 1) % a.ckpt_every == 0
                if last_iter or evaluating:               # whoever reads Agent.model needs the pass finished
                    m.train_join()
            else:
                losses = m.train(m.N_EPOCHS, it)
        except ValueError as e:                           # replay still smaller than BATCH_SIZE x TRAIN_STEPS
            losses = []
            emit(dict(kind="note", iter=it, msg="train skipped: %s" % e))
        torch.cuda.synchronize()
        t_tr = time.time() - t0
        games_total += out["episodes"]
        moves_total += out["moves"]
        ev, evg = eng.fp16_range_events()
        rec = dict(kind="iter", iter=it, games=out["episodes"], moves=out["moves"], self_play_s=round(t_sp, 2),
                   moves_per_s=round(out["moves"] / t_sp, 1), self_play_phases_s=ph, mean_game_len=round(lengths, 2), result=res,
                   train_s=round(t_tr, 2), steps=len(losses), opt_step=m.step,
                   loss=[round(float(x), 4) for x in np.mean(np.array(losses), axis=0)] if losses else None,
                   mean_select_depth=round(d["levels"] / max(d["evaluated"] + d["terminal"], 1), 3),
                   terminal_share=round(d["terminal"] / max(d["evaluated"] + d["terminal"], 1), 4),
                   trims=dict(m.trim_stats), node_cap=eng.node_cap()[0], fp16_range_events=ev, slots=eng.G, rows=eng.row_stats(),
                   skipped_steps=m.skipped_steps, mfma_products=(m._evaluator._net.products()[0] if m._evaluator._net is not None else None),
                   replay=len(m.rep_memory), games_total=games_total, moves_total=moves_total)
        emit(rec)
        m.reset_iter(m.result, m.cur_memory)
        it += 1
        if it % a.eval_every == 0 or it <= a.eval_dense_until:
            t0 = time.time()
            m.Agent.model.eval()
            if a.yardstick:
                kind, ysims = a.yardstick.split(":")
                yr, (ype, yee), yg = evaluate.evaluate_batched(m.Agent.model, kind, a.board, a.eval_sims or a.sims, n_mcts_enemy=int(ysims),
                                                               n_match=a.eval_matches, seed=9000 + it, device=0)
                ysc = (yr["Player"] + 0.5 * yr["Draw"]) / max(sum(yr.values()), 1)
                emit(dict(kind="elo", iter=it, vs=a.yardstick, matches=a.eval_matches, result=yr, elo_gain_reference_K32=round(ype - 1500.0, 1),
                          score=round(ysc, 4), mean_plies=round(float(np.mean([len(g[1]) for g in yg])), 1), eval_s=round(time.time() - t0, 2)))
                t0 = time.time()
            result, (pe, ee), games = evaluate.evaluate_batched(m.Agent.model, base, a.board, a.eval_sims or a.sims,
                                                                n_match=a.eval_matches, seed=1000 + it, device=0)
            w, l, dr = result["Player"], result["Enemy"], result["Draw"]
            score = (w + 0.5 * dr) / max(w + l + dr, 1)
            ml = 400.0 * np.log10(max(score, 1e-3) / max(1 - score, 1e-3))
            emit(dict(kind="elo", iter=it, vs="iter0", matches=a.eval_matches, sims=a.eval_sims or a.sims, result=result,
                      player_elo=round(pe, 1), enemy_elo=round(ee, 1), elo_gain_reference_K32=round(pe - 1500.0, 1),
                      score=round(score, 4), elo_diff_from_score=round(float(ml), 1),
                      mean_plies=round(float(np.mean([len(g[1]) for g in games])), 1), eval_s=round(time.time() - t0, 2)))
            # ... and against the network of the previous evaluation point: the chain of these differences keeps measuring
            # progress after "beats iteration 0 every time" has saturated
            if prev_model is not None and it % a.eval_every == 0:
                t0 = time.time()
                result, (pe, ee), games = evaluate.evaluate_batched(m.Agent.model, prev_model, a.board, a.eval_sims or a.sims,
                                                                    n_match=a.eval_matches, seed=5000 + it, device=0)
                w, l, dr = result["Player"], result["Enemy"], result["Draw"]
                score = (w + 0.5 * dr) / max(w + l + dr, 1)
                ml = 400.0 * np.log10(max(score, 0.01) / max(1 - score, 0.01))
                chain += float(ml)
                emit(dict(kind="elo", iter=it, vs="iter%d" % prev_iter, matches=a.eval_matches, result=result,
                          elo_gain_reference_K32=round(pe - 1500.0, 1), score=round(score, 4),
                          elo_diff_from_score=round(float(ml), 1), chain_elo_from_scores=round(chain, 1),
                          mean_plies=round(float(np.mean([len(g[1]) for g in games])), 1), eval_s=round(time.time() - t0, 2)))
            if it % a.eval_every == 0:
                prev_model = PVNet(a.blocks, m.IN_PLANES, a.planes, a.board).to(dev)
                prev_model.load_state_dict(m.Agent.model.state_dict())
                prev_model.eval()
                prev_iter = it
        if it % a.ckpt_every == 0:
            path = os.path.join(a.out, "ckpt_%d.pt" % it)
            torch.save(m.Agent.model.state_dict(), path)
            kept.append(path)
            while len(kept) > a.max_ckpts:                # thin out: drop the second-oldest, keep the spread
                os.remove(kept.pop(1 if len(kept) > 2 else 0))
    m.train_join()
    torch.save(m.Agent.model.state_dict(), os.path.join(a.out, "final.pt"))
    emit(dict(kind="done", iters=it, games_total=games_total, moves_total=moves_total))


if __name__ == "__main__":
    main()
