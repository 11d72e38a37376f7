"""Head-to-head evaluation of two ZeroAgents (the part of eval_main.py that sits on the hot path:
eval_main.py:137-170,191-198,229-333 without pygame / Flask).

Each player keeps its own tree; after the opponent's reply the next `get_pi(root_id)` re-roots two
plies down (known child, possibly unexpanded, or a fresh root when the reply was never reached),
exactly the call pattern of `Evaluator.get_action`. Players search with noise off and tau = 0.

`evaluate` plays the matches one after another through two drop-in agents, drawing every tie-break
from the process-global numpy stream like the reference. `evaluate_batched` plays all N_MATCH games
of eval_main.py:204-333 CONCURRENTLY on two G = n_match engines (one per network): per ply one
`ao_set_roots` launch moves every match's root in both trees and one fused search per side decides
the moves of the matches where that side is to move. Concurrent matches cannot share one stream, so
match i owns a stream seeded seed + i that both of its players draw from in turn -- match i is then
exactly `np.random.seed(seed + i); play_match(...)`.
"""
import numpy as np

from . import utils


def elo(player_elo, enemy_elo, p_winscore, e_winscore):
    """eval_main.py:191-198 (K = 32)."""
    elo_diff = enemy_elo - player_elo
    ex_pw = 1 / (1 + 10 ** (elo_diff / 400))
    ex_ew = 1 / (1 + 10 ** (-elo_diff / 400))
    player_elo += 32 * (p_winscore - ex_pw)
    enemy_elo += 32 * (e_winscore - ex_ew)
    return player_elo, enemy_elo


def _is_zero(agent):
    """ZeroAgent-style get_pi(root_id, tau) vs the rollout agents' get_pi(root_id, board, turn, tau)
    (the isinstance test of eval_main.py:139,146)."""
    inner = getattr(agent, "ag", agent)
    return not hasattr(inner, "_MODE")


def play_match(player, enemy, board_size, enemy_turn, max_plies=None, monitor=None):
    """One game; `enemy_turn` (0 black / 1 white) is the colour of `enemy` (eval_main.py:229-300). Players may be
    ZeroAgents or the rollout agents (PUCTAgent / UCTAgent); when the PLAYER is a rollout agent and a `monitor`
    ZeroAgent is given, the monitor searches the same root right after the player, as Evaluator.get_action does
    (eval_main.py:141-144: its tie-breaks come out of the shared stream). Returns (win_index, moves)."""
    win_mark = 3 if board_size == 3 else 5
    root_id = (0,)
    turn = 0
    win_index = 0
    moves = []
    while win_index == 0:
        agent = enemy if turn == enemy_turn else player
        if _is_zero(agent):
            pi = agent.get_pi(root_id, tau=0)
        else:
            pi = agent.get_pi(root_id, utils.get_board(root_id, board_size), turn, tau=0)
            if agent is player and monitor is not None:
                monitor.get_pi(root_id, tau=0)
        _, action_index = utils.argmax_onehot(pi)
        root_id = root_id + (int(action_index),)
        moves.append(int(action_index))
        win_index = utils.check_win(utils.get_board(root_id, board_size), win_mark)
        turn ^= 1
        if max_plies and len(moves) >= max_plies:
            break
    player.reset()
    enemy.reset()
    if monitor is not None:
        monitor.reset()
    return win_index, moves


def evaluate(player, enemy, board_size, n_match=12, player_elo=1500.0, enemy_elo=1500.0, return_games=False, monitor=None):
    """n_match games with the colours swapped every game (eval_main.py:213-333). Returns the result
    tally and the final ELO pair (and the [(win_index, moves)] list with return_games)."""
    result = {'Player': 0, 'Enemy': 0, 'Draw': 0}
    enemy_turn = 1
    games = []
    for _ in range(n_match):
        win_index, moves = play_match(player, enemy, board_size, enemy_turn, monitor=monitor)
        games.append((win_index, moves))
        if win_index == 3:
            result['Draw'] += 1
            pw = ew = 0.5
        elif (win_index == 1) == (enemy_turn == 1):   # black won and the player was black, or ...
            result['Player'] += 1
            pw, ew = 1.0, 0.0
        else:
            result['Enemy'] += 1
            pw, ew = 0.0, 1.0
        player_elo, enemy_elo = elo(player_elo, enemy_elo, pw, ew)
        enemy_turn ^= 1
    if return_games:
        return result, (player_elo, enemy_elo), games
    return result, (player_elo, enemy_elo)


class _ZeroSide:
    """One network's side of evaluate_batched: a G-game engine + evaluator; searches the masked matches at their ids."""

    def __init__(self, model, sims, board_size, inplanes, G, device):
        from .engine import Engine
        from .evaluator import Evaluator
        self.eng = Engine(board_size, sims, inplanes, games=G, noise=False, device=device)
        self.ev, self.model = Evaluator(device), model
        self.tau0 = np.zeros(G, np.int8)

    def seed(self, i, s):
        self.eng.seed(i, s)

    def get_rng_state(self, i):
        return self.eng.get_rng_state(i)

    def set_rng_state(self, i, mt, pos, hg, gs):
        self.eng.set_rng_state(i, mt, pos, hg, gs)

    def search(self, ids, mask):
        self.eng.set_roots(ids, mask)
        pi, _, _ = self.ev.search(self.eng, self.model, self.tau0, active=mask)
        return pi

    def close(self):
        self.eng.close()


class _RolloutSide:
    """A PUCTAgent / UCTAgent side ('puct' / 'uct', agents.py:263-614): every get_pi is a fresh search, one kernel for
    all masked matches."""

    def __init__(self, kind, sims, board_size, G, device):
        from .rollout import PUCT, UCT, RolloutEngine
        self.eng = RolloutEngine(board_size, sims, PUCT if kind == 'puct' else UCT, games=G, device=device)

    def seed(self, i, s):
        self.eng.seed(i, s)

    def get_rng_state(self, i):
        return self.eng.get_rng_state(i)

    def set_rng_state(self, i, mt, pos, hg, gs):
        self.eng.set_rng_state(i, mt, pos, hg, gs)

    def search(self, ids, mask):
        pi, _, _ = self.eng.search(ids, active=mask)
        return pi

    def close(self):
        self.eng.close()


def evaluate_batched(player_model, enemy_model, board_size, n_mcts_player, n_mcts_enemy=None, inplanes=5, n_match=12,
                     player_elo=1500.0, enemy_elo=1500.0, seed=0, device=0, max_plies=None, monitor_model=None,
                     n_mcts_monitor=None):
    """All n_match games at once (colours swapped every game, eval_main.py:213-333). `*_model`: whatever
    ZeroAgent.model accepts (a PVNet-shaped module runs on the native forward), or 'puct' / 'uct' for the rollout
    agents (eval_main.py:68-73,106-111). With a rollout PLAYER and a `monitor_model`, the monitor ZeroAgent searches
    the same roots right after the player as Evaluator.get_action does (eval_main.py:141-144). Returns (result tally,
    (player_elo, enemy_elo), [(win_index, moves) per match]); the ELO updates are applied in match order."""
    n_mcts_enemy = n_mcts_player if n_mcts_enemy is None else n_mcts_enemy
    G = n_match
    win_mark = 3 if board_size == 3 else 5

    def make(model, sims):
        if isinstance(model, str):
            return _RolloutSide(model, sims, board_size, G, device)
        return _ZeroSide(model, sims, board_size, inplanes, G, device)

    sides = [make(player_model, n_mcts_player), make(enemy_model, n_mcts_enemy)]
    monitor = None
    if monitor_model is not None and isinstance(player_model, str):
        monitor = _ZeroSide(monitor_model, n_mcts_monitor or n_mcts_enemy, board_size, inplanes, G, device)
        sides.append(monitor)                                       # index 2: searches right after side 0
    enemy_turn = np.array([1 - (i % 2) for i in range(G)])          # match 0: the player is black
    ids = [(0,) for _ in range(G)]
    wins = np.zeros(G, np.int64)
    running = np.ones(G, bool)
    # one stream per match, handed from the side that just searched to the side that searches next
    holder = np.where(enemy_turn == 0, 1, 0)                        # the side that moves first holds the stream
    for i in range(G):
        sides[holder[i]].seed(i, (seed + i) & 0xFFFFFFFF)

    def take_stream(side, mask):
        for i in np.nonzero(mask)[0]:
            if holder[i] != side:                                    # another side drew last: take the stream over
                mt, pos, hg, gs = sides[holder[i]].get_rng_state(int(i))
                sides[side].set_rng_state(int(i), mt, pos, hg, gs)
                holder[i] = side

    ply = 0
    while running.any():
        turn = ply % 2
        for side in (0, 1):                                          # 0 player, 1 enemy
            mask = running & ((enemy_turn == turn) == (side == 1))
            if not mask.any():
                continue
            m8 = mask.astype(np.uint8)
            take_stream(side, mask)
            pi = sides[side].search(ids, m8)
            if side == 0 and monitor is not None:
                take_stream(2, mask)
                monitor.search(ids, m8)
            for i in np.nonzero(mask)[0]:
                a = int(np.argmax(pi[i]))                            # argmax_onehot of a one-hot: no draw (utils.py:198-205)
                ids[i] = ids[i] + (a,)
                wins[i] = utils.check_win(utils.get_board(ids[i], board_size), win_mark)
                if wins[i] != 0 or (max_plies and len(ids[i]) - 1 >= max_plies):
                    running[i] = False
        ply += 1
    for sd in sides:
        sd.close()
    result = {'Player': 0, 'Enemy': 0, 'Draw': 0}
    games = []
    for i in range(G):
        w = int(wins[i])
        games.append((w, list(ids[i][1:])))
        if w == 3 or w == 0:
            result['Draw'] += 1
            pw = ew = 0.5
        elif (w == 1) == (enemy_turn[i] == 1):
            result['Player'] += 1
            pw, ew = 1.0, 0.0
        else:
            result['Enemy'] += 1
            pw, ew = 0.0, 1.0
        player_elo, enemy_elo = elo(player_elo, enemy_elo, pw, ew)
    return result, (player_elo, enemy_elo), games
