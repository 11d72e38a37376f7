"""Head-to-head evaluation of two ZeroAgents (the part of eval_main.py that sits on the hot path:
eval_main.py:137-170,191-198,229-333 without pygame / Flask).

Each player keeps its own tree; after the opponent's reply the next `get_pi(root_id)` re-roots two
plies down (known child, possibly unexpanded, or a fresh root when the reply was never reached),
exactly the call pattern of `Evaluator.get_action`. Players search with noise off and tau = 0.
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


def play_match(player, enemy, board_size, enemy_turn, max_plies=None):
    """One game; `enemy_turn` (0 black / 1 white) is the colour of `enemy` (eval_main.py:229-300).
    Returns (win_index, moves)."""
    win_mark = 3 if board_size == 3 else 5
    root_id = (0,)
    turn = 0
    win_index = 0
    moves = []
    while win_index == 0:
        agent = enemy if turn == enemy_turn else player
        pi = agent.get_pi(root_id, tau=0)
        _, action_index = utils.argmax_onehot(pi)
        root_id = root_id + (int(action_index),)
        moves.append(int(action_index))
        win_index = utils.check_win(utils.get_board(root_id, board_size), win_mark)
        turn ^= 1
        if max_plies and len(moves) >= max_plies:
            break
    player.reset()
    enemy.reset()
    return win_index, moves


def evaluate(player, enemy, board_size, n_match=12, player_elo=1500.0, enemy_elo=1500.0):
    """n_match games with the colours swapped every game (eval_main.py:213-333). Returns the result
    tally and the final ELO pair."""
    result = {'Player': 0, 'Enemy': 0, 'Draw': 0}
    enemy_turn = 1
    for _ in range(n_match):
        win_index, _ = play_match(player, enemy, board_size, enemy_turn)
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
    return result, (player_elo, enemy_elo)
