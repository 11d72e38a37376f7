#!/usr/bin/env python3
"""Numerics gate for the 3-byte activation format of the split-fp16 trunk (round-2 review item 2), emulated in torch
(CPU, float64 convolutions) on the golden-vector networks (tests/pvnet_weights.py: non-trivial BatchNorm statistics)
-- no kernel involved.

Shipped format ("split3"): every trunk activation is stored as two fp16 halves, x = xh + xl, 4 bytes, ~22 significand bits.
Candidate ("lo8"): activations are post-ReLU, so x >= 0; store the fp16 TRUNCATION of x (2 bytes: 5 exponent + 10 mantissa
bits) and the NEXT 8 mantissa bits in one byte -- 3 bytes, 19 significand bits.  On the device that is integer work:
    t = (bits(x) - (112 << 23)) >> 5       # 24 bits: [E5 | M18]; hi16 = t >> 8 (a valid fp16), lo8 = t & 255
and the low half the MFMA needs is xl = lo8 * 2^(E5 - 15 - 18), built as (base | lo8 << 2) - base with
base = fp16 bits ((E5 - 10) << 10), which needs E5 >= 11 (x >= 2^-4 after the activation pre-scale 2^k that the
BatchNorm scale absorbs); below that the low byte is ignored ("flush").  The contraction itself is unchanged:
x*w ~ xh*wh + xh*wl + xl*wh with fp32 accumulation.

Variants printed:
  split3          the shipped path
  lo8 k=K         the candidate with activation pre-scale 2^K (K = 0, 4), low byte ignored below 2^-4 / 2^K
  lo8 k=K exact   same, low byte honoured down to fp16's subnormal quantum (what an fp32-path conversion would give)
  lo_bf8          low half stored as fp8 e5m2 (the top byte of the fp16 low half: a byte shift to convert)
  hi_only         no low half at all

    python tools/emulate_lo8_storage.py     # max |dp|, |dv| against the exact (fp64) evaluation; bar 1e-4, gate 2e-5
"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import pvnet_weights

torch.set_num_threads(8)


def f16_trunc(x):
    """fp16 round-toward-zero of non-negative float64 x (normal range; below 2^-14: fp16 subnormal grid)."""
    e = torch.floor(torch.log2(torch.clamp(x, min=2.0 ** -30)))
    e = torch.clamp(e, min=-14.0)
    q = torch.pow(2.0, e - 10.0)
    return torch.floor(x / q) * q


def quant_act(x, fmt, k):
    """What the epilogue stores and the next layer's MFMAs see: returns (xh, xl) as float64 tensors."""
    if fmt == "split3":
        xh = x.to(torch.float16).to(torch.float64)
        xl = (x - xh).to(torch.float16).to(torch.float64)
        return xh, xl
    if fmt == "hi_only":
        return x.to(torch.float16).to(torch.float64), torch.zeros_like(x)
    if fmt == "lo_bf8":
        xh = x.to(torch.float16).to(torch.float64)
        xl = (x - xh).to(torch.float32).to(torch.float8_e5m2).to(torch.float64)
        return xh, xl
    s = 2.0 ** k
    xs = torch.clamp(x * s, max=65504.0)
    if fmt.endswith("r"):   # round to nearest on the 19-bit grid first (the device adds half a step to the integer before the shift)
        e0 = torch.clamp(torch.floor(torch.log2(torch.clamp(xs, min=2.0 ** -30))), min=-14.0)
        q0 = torch.pow(2.0, e0 - 18.0)
        xs = torch.clamp(torch.floor(xs / q0 + 0.5) * q0, max=65504.0)
        fmt = fmt[:-1]
    xh = f16_trunc(xs)
    e = torch.floor(torch.log2(torch.clamp(xh, min=2.0 ** -30)))
    q = torch.pow(2.0, e - 18.0)
    lo = torch.floor((xs - xh) / q)  # 0..255
    lo = torch.clamp(lo, 0, 255)
    xl = lo * q
    if fmt == "lo8":  # packed-fp16 reconstruction: only for E5 >= 11
        xl = torch.where(e >= -4.0, xl, torch.zeros_like(xl))
    else:             # "lo8x": fp16 subnormal quantum 2^-24 is the floor
        xl = torch.floor(xl * 2.0 ** 24) / 2.0 ** 24
    xl = torch.where(xh >= 2.0 ** -14, xl, torch.zeros_like(xl))
    return xh / s, xl / s


def split_w(w):
    mx = w.abs().max().item()
    s = 2.0 ** (2 - int(np.floor(np.log2(mx)))) if mx > 0 else 1.0
    wh = (w * s).to(torch.float16).to(torch.float64)
    wl = (w * s - wh).to(torch.float16).to(torch.float64)
    return wh, wl, s


def conv(xh, xl, w, exact):
    pad = w.shape[-1] // 2
    if exact:
        return F.conv2d(xh + xl, w, padding=pad)
    wh, wl, s = split_w(w)
    return (F.conv2d(xh, wh, padding=pad) + F.conv2d(xh, wl, padding=pad) + F.conv2d(xl, wh, padding=pad)) / s


def bn(x, sd, pre):
    w, b, m, v = (torch.from_numpy(sd[pre + "." + k]).double() for k in ("weight", "bias", "running_mean", "running_var"))
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-5) * w[None, :, None, None] + b[None, :, None, None]


def forward(sd, x, nb, fmt, k=0):
    t = lambda n: torch.from_numpy(sd[n]).double()
    exact = fmt == "exact"
    q = (lambda a: (a, torch.zeros_like(a))) if exact else (lambda a: quant_act(a, fmt, k))
    # conv1: 0/1 planes, exact in fp16
    h = q(F.relu(bn(conv(x, torch.zeros_like(x), t("conv1.weight"), exact), sd, "bn1").float().double() if not exact
                 else bn(conv(x, torch.zeros_like(x), t("conv1.weight"), exact), sd, "bn1")))
    f32 = (lambda a: a) if exact else (lambda a: a.float().double())   # the epilogue works in fp32
    for i in range(nb):
        p = "layers.%d." % i
        y = q(F.relu(f32(bn(conv(h[0], h[1], t(p + "conv1.weight"), exact), sd, p + "bn1"))))
        h = q(F.relu(f32(bn(conv(y[0], y[1], t(p + "conv2.weight"), exact), sd, p + "bn2") + (h[0] + h[1]))))
    h = h[0] + h[1]
    ph = F.relu(bn(F.conv2d(h, t("policy_head.policy_head.weight")), sd, "policy_head.policy_bn")).flatten(1)
    pol = F.softmax(ph @ t("policy_head.policy_fc.weight").T + t("policy_head.policy_fc.bias"), dim=-1)
    vh = F.relu(bn(F.conv2d(h, t("value_head.value_head.weight")), sd, "value_head.value_bn")).flatten(1)
    v1 = F.relu(vh @ t("value_head.value_fc1.weight").T + t("value_head.value_fc1.bias"))
    val = torch.tanh(v1 @ t("value_head.value_fc2.weight").T + t("value_head.value_fc2.bias")).squeeze(-1)
    return pol, val, h


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    print("%-28s %-16s %12s %12s" % ("network", "variant", "max |dp|", "max |dv|"))
    nets = [(4, 9, 77), (4, 9, 3), (10, 9, 5), (10, 15, 8)]
    if "--default-init" in sys.argv:
        nets = []
    for nb, B, seed in nets:
        sd = pvnet_weights.make_state_dict(nb, 5, 128, B, seed)
        x = torch.from_numpy((rs.rand(24, 5, B, B) < 0.3).astype(np.float64))
        p0, v0, h0 = forward(sd, x, nb, "exact")
        print("# trunk output: max %.3g, median of non-zeros %.3g, share below 2^-4: %.3f" % (
            h0.max().item(), h0[h0 > 0].median().item(), ((h0 > 0) & (h0 < 2.0 ** -4)).double().mean().item()))
        for fmt, k in (("split3", 0), ("lo8", 0), ("lo8", 4), ("lo8r", 0), ("lo8r", 4), ("lo8xr", 4), ("lo_bf8", 0), ("hi_only", 0)):
            p, v, _ = forward(sd, x, nb, fmt, k)
            name = fmt if fmt in ("split3", "lo_bf8", "hi_only") else "%s k=%d" % (fmt, k)
            print("%-28s %-16s %12.2e %12.2e" % ("%d blocks, %dx%d, seed %d" % (nb, B, B, seed), name,
                                                 (p - p0).abs().max().item(), (v - v0).abs().max().item()))
    # the bench's network: PyTorch default init, BatchNorm gamma 1 / beta 0 (model.py:86-89)
    from alpha_omok_amd.pvnet import PVNet
    for nb, B in ((4, 9), (10, 9), (10, 15)):
        torch.manual_seed(0)
        m = PVNet(nb, 5, 128, B).eval()
        sd = {k_: v_.numpy() for k_, v_ in m.state_dict().items()}
        x = torch.from_numpy((rs.rand(24, 5, B, B) < 0.3).astype(np.float64))
        p0, v0, h0 = forward(sd, x, nb, "exact")
        print("# trunk output: max %.3g, median of non-zeros %.3g, share below 2^-4: %.3f" % (
            h0.max().item(), h0[h0 > 0].median().item(), ((h0 > 0) & (h0 < 2.0 ** -4)).double().mean().item()))
        for fmt, k in (("split3", 0), ("lo8", 0), ("lo8r", 0), ("lo8r", 4), ("lo8xr", 4), ("lo_bf8", 0)):
            p, v, _ = forward(sd, x, nb, fmt, k)
            name = fmt if fmt in ("split3", "lo_bf8", "hi_only") else "%s k=%d" % (fmt, k)
            print("%-28s %-16s %12.2e %12.2e" % ("%d blocks, %dx%d, default init" % (nb, B, B), name,
                                                 (p - p0).abs().max().item(), (v - v0).abs().max().item()))
