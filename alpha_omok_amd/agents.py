"""Drop-in `agents.ZeroAgent` (reference: agents.py:16-260) and the net-free rollout agents
`PUCTAgent` / `UCTAgent` (agents.py:263-614), all backed by the HIP engine.

Same constructor, attributes and methods as the reference class; the per-simulation Python loop
is replaced by the batched tree kernels with G = 1. The evaluator is `self.model`, assigned after
construction exactly as main.py:81 / eval_main.py:87-101 do:
  * a module whose state_dict has the PVNet wire format runs on the hand-written MFMA forward
    (weights are re-exported whenever the module's parameters change);
  * any other callable `model(x[B,C,Bd,Bd]) -> (policy[B,A], value[B])` is called once per
    simulation on the leaf batch (planes are produced on the device).
Randomness is the process-global numpy stream (main.py:60): its MT19937 state is moved into the
engine for the search and written back afterwards, so `np.random.seed(s)` reproduces the
reference's visit counts and the stream position it leaves behind.
"""
import sys
import time

import numpy as np

from .evaluator import Evaluator
from .engine import AO_ROOT_FRESH, Engine, EngineError  # noqa: F401

PRINT_MCTS = True


class Agent(object):
    def __init__(self, board_size):
        self.policy = np.zeros(board_size ** 2, 'float')
        self.visit = np.zeros(board_size ** 2, 'float')
        self.message = 'Hello'

    def get_policy(self):
        return self.policy

    def get_visit(self):
        return self.visit

    def get_name(self):
        return type(self).__name__

    def get_message(self):
        return self.message

    def get_pv(self, root_id):
        return None, None


class _TreeView(object):
    """Stands in for the reference's `self.tree` dict where callers only take len() / clear()."""

    def __init__(self, agent):
        self._agent = agent

    def __len__(self):
        eng = self._agent._engine
        return 0 if eng is None else eng.tree_nodes(0)[1]

    def __bool__(self):
        return len(self) > 0

    def clear(self):
        if self._agent._engine is not None:
            self._agent._engine.reset()


class ZeroAgent(Agent):
    def __init__(self, board_size, num_mcts, inplanes, noise=True, device=0, node_cap=0):
        super(ZeroAgent, self).__init__(board_size)
        self.board_size = board_size
        self.num_mcts = num_mcts
        self.inplanes = inplanes
        self.win_mark = 3 if board_size == 3 else 5
        self.alpha = 10 / self.board_size ** 2
        self.c_puct = 5
        self.noise = noise
        self.root_id = None
        self.model = None
        self.is_real_root = True
        self.tree = _TreeView(self)
        self._device = device
        self._node_cap = node_cap
        self._engine = None
        self._evaluator = Evaluator(device)

    # -- engine plumbing --------------------------------------------------------------------
    def _eng(self):
        if self._engine is None:
            self._engine = Engine(self.board_size, self.num_mcts, self.inplanes, games=1, noise=self.noise,
                                  device=self._device, node_cap=self._node_cap, c_puct=float(self.c_puct),
                                  alpha=float(self.alpha), win_mark=self.win_mark)
        return self._engine

    # -- reference API ----------------------------------------------------------------------
    def reset(self):
        self.root_id = None
        self.is_real_root = True
        if self._engine is not None:
            self._engine.reset()

    def get_pi(self, root_id, tau):
        start = time.time()
        eng = self._eng()
        st = np.random.get_state()                      # the reference's global stream
        eng.set_rng_state(0, st[1], st[2], st[3], st[4])
        status = eng.set_root(0, list(root_id)[1:])
        self.root_id = tuple(root_id)
        self.is_real_root = (status == AO_ROOT_FRESH)
        num = self.num_mcts + 1 if self.is_real_root else self.num_mcts
        def progress(i):
            self.message = 'simulation: {}\r'.format(i)

        pi, visit, policy = self._evaluator.search(eng, self.model, tau, on_sim=progress)
        mt, pos, has_gauss, gauss = eng.get_rng_state(0)
        np.random.set_state(('MT19937', mt, pos, has_gauss, gauss))
        self.message = 'simulation: {}\r'.format(num)
        self.visit = visit[0]
        self.policy = policy[0]
        if PRINT_MCTS:
            sys.stdout.write('simulation: {}\r'.format(num))
            print("{} simulations end ({:0.0f}s)".format(num, time.time() - start))
        return pi[0]

    def del_parents(self, root_id):
        """The engine keeps only the subtree of the last root (what del_parents leaves reachable)."""
        expanded, entries = (0, 0) if self._engine is None else self._engine.tree_nodes(0)
        print('tree size:', entries)
        print('tree depth:', 0 if expanded == 0 else '>= 1')

    def get_pv(self, root_id):
        import torch
        from . import utils
        state = utils.get_state_pt(root_id, self.board_size, self.inplanes)
        x = torch.from_numpy(state[None]).float()
        net = self._evaluator.native_net(self.model, self.board_size, self.inplanes)
        with torch.no_grad():
            if net is not None:
                p, v = net(x.cuda(self._device))
            else:
                if hasattr(self.model, "eval"):
                    self.model.eval()
                p, v = self.model(x.to(Evaluator._model_device(self.model)))
        return p.detach().cpu().numpy()[0], v.detach().cpu().numpy()[0]


class _RolloutAgent(Agent):
    """Shared body of PUCTAgent / UCTAgent (agents.py:263-614): every get_pi is a fresh search of
    num_mcts + 1 simulations with random playouts, run as ONE kernel on the device with the
    process-global np.random state (moved in and out as ZeroAgent does)."""
    _MODE = 0

    def __init__(self, board_size, num_mcts, device=0):
        super(_RolloutAgent, self).__init__(board_size)
        self.board_size = board_size
        self.num_mcts = num_mcts
        self.win_mark = 3 if board_size == 3 else 5
        self.root_id = None
        self.board = None
        self.turn = None
        self.is_real_root = True
        self._device = device
        self._engine = None
        self.tree = {}     # the reference keeps its dict across calls but never reads old entries

    def reset(self):
        self.is_real_root = True
        self.root_id = None
        self.board = None
        self.turn = None
        self.tree.clear()

    def get_pi(self, root_id, board, turn, tau):
        from . import utils
        from .rollout import RolloutEngine
        if turn != utils.get_turn(root_id):
            raise ValueError("turn does not match root_id")
        if self._engine is None:
            self._engine = RolloutEngine(self.board_size, self.num_mcts, self._MODE, games=1, device=self._device)
        self.root_id, self.board, self.turn = root_id, board, turn
        start = time.time()
        st = np.random.get_state()
        self._engine.set_rng_state(0, st[1], st[2], st[3], st[4])
        pi, stat, _ = self._engine.search([root_id])
        mt, pos, hg, gs = self._engine.get_rng_state(0)
        np.random.set_state(('MT19937', mt, pos, hg, gs))
        if self._MODE == 0:
            self.visit = stat[0].copy()
        self.tree = {root_id: {'child': [a for a in range(self.board_size ** 2) if a not in root_id[1:]]}}
        if PRINT_MCTS:
            print("{} simulations end ({:0.0f}s)".format(self.num_mcts, time.time() - start))
        return pi[0].copy()

    def del_parents(self, root_id):
        self.tree = {k: v for k, v in self.tree.items() if len(k) >= len(root_id)}


class PUCTAgent(_RolloutAgent):
    """agents.py:263-441: PUCT over uniform priors, value from one random playout per expansion."""
    _MODE = 0


class UCTAgent(_RolloutAgent):
    """agents.py:443-614: UCB1 (unvisited children first), value from one random playout per expansion."""
    _MODE = 1
