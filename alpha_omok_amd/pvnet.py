"""Policy/value ResNet as a torch module, for training and as the fp32 reference of the HIP
forward (alpha_omok_amd/csrc/net.hip).

State-dict compatible with the reference's ``model.PVNet`` (model.py:76-104): identical key
names and shapes, so checkpoints interchange. The attribute names
below are therefore fixed by that wire format; everything else is this package's own code.
Inference in the engine does not go through this module: `to_native()` exports the weights to
the hand-written MFMA kernels.
"""
import torch
import torch.nn.functional as F
from torch import nn


def _conv(cin, cout, k):
    return nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=False)


class _Block(nn.Module):
    """Two 3x3 conv+BN with an identity skip (model.py:13-31)."""

    def __init__(self, ch):
        super().__init__()
        self.conv1, self.bn1 = _conv(ch, ch, 3), nn.BatchNorm2d(ch)
        self.conv2, self.bn2 = _conv(ch, ch, 3), nn.BatchNorm2d(ch)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + x)


class _Policy(nn.Module):
    def __init__(self, ch, cells):
        super().__init__()
        self.policy_head, self.policy_bn = _conv(ch, 2, 1), nn.BatchNorm2d(2)
        self.policy_fc = nn.Linear(2 * cells, cells)

    def forward(self, x):
        h = F.relu(self.policy_bn(self.policy_head(x))).flatten(1)  # NCHW flatten (model.py:47)
        return F.softmax(self.policy_fc(h), dim=-1)


class _Value(nn.Module):
    def __init__(self, ch, cells):
        super().__init__()
        self.value_head, self.value_bn = _conv(ch, 1, 1), nn.BatchNorm2d(1)
        self.value_fc1, self.value_fc2 = nn.Linear(cells, ch), nn.Linear(ch, 1)

    def forward(self, x):
        h = F.relu(self.value_bn(self.value_head(x))).flatten(1)
        return torch.tanh(self.value_fc2(F.relu(self.value_fc1(h)))).squeeze(-1)


class PVNet(nn.Module):
    """PVNet(n_block, inplanes, planes, board_size): same constructor as the reference."""

    def __init__(self, n_block, inplanes, planes, board_size):
        super().__init__()
        self.cfg = (n_block, inplanes, planes, board_size)
        cells = board_size * board_size
        self.conv1, self.bn1 = _conv(inplanes, planes, 3), nn.BatchNorm2d(planes)
        self.layers = nn.Sequential(*[_Block(planes) for _ in range(n_block)])
        self.policy_head = _Policy(planes, cells)
        self.value_head = _Value(planes, cells)
        for m in self.modules():  # model.py:86-89
            if isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        t = self.layers(F.relu(self.bn1(self.conv1(x))))
        return self.policy_head(t), self.value_head(t)

    def to_native(self, device=0, net=None):
        """Export the current weights to the HIP forward. Returns an alpha_omok_amd.engine.Net. Widths that are not a multiple of
        32 (model.py:76-85 takes any `planes`) are exported zero-padded to the next multiple (`pad_state_dict`): same function."""
        from .engine import Net
        n_block, inplanes, planes, board_size = self.cfg
        width = native_width(planes)
        if net is None:
            net = Net(n_block, inplanes, width, board_size, device)
        sd = self.state_dict()
        net.load_state_dict(sd if width == planes else pad_state_dict(sd, width))
        return net


NATIVE_MAX_PLANES = 512


def native_supported(planes):
    """True when the HIP forward takes a `planes`-wide network (zero-padded to the next multiple of 32 if need be)."""
    return 1 <= int(planes) <= NATIVE_MAX_PLANES


def native_width(planes):
    """The width the HIP forward runs a `planes`-wide network at: the next multiple of 32 (at most NATIVE_MAX_PLANES = 512), or
    `planes` itself when the native kernels cannot take it (wider: the caller's torch module evaluates it)."""
    width = (int(planes) + 31) // 32 * 32
    return width if 32 <= width <= NATIVE_MAX_PLANES else int(planes)


def pad_state_dict(state_dict, width):
    """PVNet state_dict of `planes` channels -> the state_dict of the SAME function at `width` >= planes channels: the extra channels
    have zero conv weights and identity BatchNorm statistics (mean 0, variance 1, shift 0), so they carry exact zeros through the
    ReLUs and the skips; the 1x1 head convs, value_fc1 (its hidden width is `planes` too) and value_fc2 get zero columns / rows
    for them. Every added term of every sum is +0.0: the outputs are those of the unpadded network up to the order of summation."""
    import numpy as np
    out = {}
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        name = k.split(".")[-1]
        if k.endswith("conv1.weight") and k == "conv1.weight":
            p = np.zeros((width,) + a.shape[1:], a.dtype); p[:a.shape[0]] = a
        elif name == "weight" and a.ndim == 4 and a.shape[2] == 3:                 # trunk convs [planes, planes, 3, 3]
            p = np.zeros((width, width) + a.shape[2:], a.dtype); p[:a.shape[0], :a.shape[1]] = a
        elif name == "weight" and a.ndim == 4:                                    # 1x1 head convs [2 | 1, planes, 1, 1]
            p = np.zeros((a.shape[0], width) + a.shape[2:], a.dtype); p[:, :a.shape[1]] = a
        elif k == "value_head.value_fc1.weight":                                  # [planes, cells]
            p = np.zeros((width, a.shape[1]), a.dtype); p[:a.shape[0]] = a
        elif k == "value_head.value_fc1.bias":
            p = np.zeros((width,), a.dtype); p[:a.shape[0]] = a
        elif k == "value_head.value_fc2.weight":                                  # [1, planes]
            p = np.zeros((1, width), a.dtype); p[:, :a.shape[1]] = a
        elif a.ndim == 1 and name in ("weight", "bias", "running_mean", "running_var") and "policy_bn" not in k and "value_bn" not in k \
                and "fc" not in k:
            fill = 1.0 if name in ("weight", "running_var") else 0.0              # trunk BatchNorms [planes]
            p = np.full((width,), fill, a.dtype); p[:a.shape[0]] = a
        else:
            p = a
        out[k] = p
    return out


def looks_like_pvnet(module):
    """(n_block, inplanes, planes, board) if `module`'s state_dict has the PVNet wire format."""
    try:
        sd = module.state_dict()
        w = sd["conv1.weight"]
        planes, inplanes = int(w.shape[0]), int(w.shape[1])
        n_block = 0
        while "layers.%d.conv1.weight" % n_block in sd:
            n_block += 1
        cells = int(sd["policy_head.policy_fc.weight"].shape[0])
        board = int(round(cells ** 0.5))
        if board * board != cells or "value_head.value_fc2.weight" not in sd:
            return None
        return n_block, inplanes, planes, board
    except Exception:
        return None
