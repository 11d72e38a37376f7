"""Runs one move decision of an Engine with whatever `Agent.model` is (agents.py:171-178).

  * PVNet-shaped torch module (state_dict wire format of model.py) -> exported to the native MFMA
    forward (engine.Net); the whole move is one fused C call (ao_search), no host round trips;
  * anything else callable as model(x[G,C,B,B]) -> (policy[G,A], value[G]) -> stepwise protocol,
    one call per simulation on the whole leaf batch.
"""
import weakref

import numpy as np

from . import pvnet


class Evaluator:
    def __init__(self, device=0):
        self.device = device
        self._key = None
        self._ref = None
        self._cfg = None
        self._net = None
        self._bufs = None
        self._warned_width = False
        self._frozen = False   # freeze(): searches keep the weights exported last, whatever happens to the module (main.train_async)
        self.strict_native = False   # True: a PVNet the native forward cannot take raises instead of running on its torch module
        self.net_mode = 0      # ao_net_set_mode of the exported network (6: one kernel family for every batch size)

    def native_net(self, model, board_size, inplanes):
        if self._frozen and self._net is not None:
            return self._net
        cfg = pvnet.looks_like_pvnet(model)
        if cfg is None or cfg[3] != board_size or cfg[1] != inplanes:
            return None
        if not pvnet.native_supported(cfg[2]):
            # model.PVNet takes any `planes` (model.py:76-85); the hand-written forward covers up to 512 (other widths zero-padded to
            # the next multiple of 32, pvnet.pad_state_dict). Wider networks: INTEGRATION.md, "Network widths".
            if self.strict_native:
                raise ValueError("PVNet with %d planes has no native MI355X forward (up to %d planes) and strict_native is set"
                                 % (cfg[2], pvnet.NATIVE_MAX_PLANES))
            if not self._warned_width:
                import warnings
                warnings.warn("PVNet with %d planes: the native MI355X forward covers up to 512 planes -- this network "
                              "is evaluated by its own torch module, one call per simulation on the whole leaf batch "
                              "(correct, but several times slower than the MFMA kernels)" % cfg[2], RuntimeWarning, stacklevel=3)
                self._warned_width = True
            return None
        width = pvnet.native_width(cfg[2])
        # the native copy is keyed on the module OBJECT (held through a weak reference: a new module
        # at a recycled address is a different object), its shape and the in-place version counters of
        # its tensors; invalidate() forces a re-export for anything those cannot see
        version = tuple(int(getattr(p, "_version", 0)) for p in model.state_dict().values())
        same_module = self._ref is not None and self._ref() is model
        if not same_module or self._key != (cfg, version):
            from .engine import Net
            if self._net is None or self._cfg != cfg:
                if self._net is not None:
                    self._net.close()
                self._net = Net(cfg[0], cfg[1], width, cfg[3], self.device)
                self._cfg = cfg
            sd = model.state_dict()
            self._net.load_state_dict(sd if width == cfg[2] else pvnet.pad_state_dict(sd, width))
            self._net.set_mode(self.net_mode)             # after every export, 0 included: an fp16-range fallback of the OLD weights ends here
            self._key = (cfg, version)
            try:
                self._ref = weakref.ref(model)
            except TypeError:
                self._ref = None
        return self._net

    def freeze(self, model, board_size, inplanes):
        """Export `model` now (if it changed) and keep searching with THAT copy until thaw(): the module's parameters may then be
        updated in place by another thread (main.train_async) -- the native forward reads its own repacked weights in HBM, never the
        module's tensors. False when the model has no native form (every simulation would call the module itself)."""
        self._frozen = False
        if self.native_net(model, board_size, inplanes) is None:
            return False
        self._frozen = True
        return True

    def thaw(self):
        """End of freeze(): the next search re-exports the module's weights."""
        self._frozen = False
        self.invalidate()

    def invalidate(self):
        """Forget the exported weights: the next search re-exports Agent.model (called by main.train and
        main.load_data; call it after replacing parameters in a way torch's version counters miss)."""
        self._key = None
        self._ref = None

    @staticmethod
    def _model_device(model):
        import torch
        try:
            return next(model.parameters()).device
        except Exception:
            return torch.device("cpu")

    def search(self, eng, model, tau, active=None, on_sim=None):
        """Returns (pi, visit, policy) float64 [G, A]."""
        tau = np.ascontiguousarray(np.broadcast_to(tau, (eng.G,)), np.int8)
        net = model if hasattr(model, "forward_ptr") else self.native_net(model, eng.board_size, eng.inplanes)
        if net is not None:
            return eng.search(net, tau=tau, active=active)
        import torch
        dev = torch.device("cuda", self.device)
        G, C, B, A = eng.G, eng.inplanes, eng.board_size, eng.A
        if self._bufs is None or tuple(self._bufs[0].shape) != (G, C, B, B):
            self._bufs = (torch.zeros((G, C, B, B), dtype=torch.float32, device=dev),
                          torch.zeros((G, A), dtype=torch.float32, device=dev),
                          torch.zeros((G,), dtype=torch.float32, device=dev))
        planes, pol, val = self._bufs
        mdev = self._model_device(model)
        if hasattr(model, "eval"):
            model.eval()
        eng.begin_move(active)
        i = 0
        while eng.sims_left() > 0:
            i += 1
            if on_sim is not None:
                on_sim(i)
            eng.collect_leaves(planes.data_ptr())
            eng.sync()
            with torch.no_grad():
                p, v = model(planes.to(mdev))
            pol.copy_(p.reshape(G, A).to(dev, torch.float32))
            val.copy_(v.reshape(G).to(dev, torch.float32))
            torch.cuda.synchronize(dev)
            eng.apply_evals(pol.data_ptr(), val.data_ptr())
        return eng.end_move(tau)
