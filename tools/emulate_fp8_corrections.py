#!/usr/bin/env python3
"""Numerics of the two cheaper trunk variants the round-1 review asked about, emulated in torch (CPU, float64 convolutions)
on the golden-vector networks (tests/pvnet_weights.py: non-trivial BatchNorm statistics) -- no kernel involved:

  split3   what k_trunk16h computes: x*w ~ xh*wh + xh*wl + xl*wh, halves in fp16            (the shipped path)
  fp8corr  the two correction operands (xl, wl) rounded to fp8 e4m3 (the 2x-rate MFMA pipe; -33 % matrix energy,
           -25 % activation bytes if the low halves are also STORED in 8 bits)
  hi_only  no correction products at all (x*w ~ xh*wh)

    python tools/emulate_fp8_corrections.py        # prints max |dp|, |dv| against the fp64 evaluation of the same network
The bar is BASELINE.json's 1e-4 absolute on policy and value."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import pvnet_weights

torch.set_num_threads(8)


def split(t, lo_dtype):
    hi = t.to(torch.float16).to(torch.float64)
    lo = (t - hi)
    if lo_dtype == "fp16":
        lo = lo.to(torch.float16).to(torch.float64)
    elif lo_dtype == "fp8":
        # the low half of a value near 2^e is below 2^(e-11): scaled by a power of two so that the low halves of the
        # LARGEST values land at the top of e4m3's range (448), the way a kernel would fold the scale into BatchNorm
        top = lo.abs().max().item()
        k = 2.0 ** (8 - int(np.ceil(np.log2(top)))) if top > 0 else 1.0
        lo = (lo * k).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) / k
    else:
        lo = torch.zeros_like(lo)
    return hi, lo


def conv(x, w, mode):
    if mode == "exact":
        return F.conv2d(x, w, padding=w.shape[-1] // 2)
    # weights are pre-scaled by a power of two per layer so that their low halves are normal numbers (as the kernel does)
    mx = w.abs().max().item()
    s = 2.0 ** (2 - int(np.floor(np.log2(mx)))) if mx > 0 else 1.0
    lo_dtype = {"split3": "fp16", "fp8corr": "fp8", "hi_only": "none"}[mode]
    xh, xl = split(x, lo_dtype)
    wh, wl = split(w * s, lo_dtype)
    pad = w.shape[-1] // 2
    y = F.conv2d(xh, wh, padding=pad)
    if mode != "hi_only":
        y = y + F.conv2d(xh, wl, padding=pad) + F.conv2d(xl, wh, padding=pad)
    return y / s


def bn(x, sd, pre):
    w, b, m, v = (torch.from_numpy(sd[pre + "." + k]).double() for k in ("weight", "bias", "running_mean", "running_var"))
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-5) * w[None, :, None, None] + b[None, :, None, None]


def forward(sd, x, nb, mode):
    t = lambda k: torch.from_numpy(sd[k]).double()
    h = F.relu(bn(conv(x, t("conv1.weight"), mode), sd, "bn1"))
    for i in range(nb):
        p = "layers.%d." % i
        y = F.relu(bn(conv(h, t(p + "conv1.weight"), mode), sd, p + "bn1"))
        h = F.relu(bn(conv(y, t(p + "conv2.weight"), mode), sd, p + "bn2") + h)
    # heads in exact arithmetic (they are a small part and are not the question here)
    ph = F.relu(bn(F.conv2d(h, t("policy_head.policy_head.weight")), sd, "policy_head.policy_bn")).flatten(1)
    pol = F.softmax(ph @ t("policy_head.policy_fc.weight").T + t("policy_head.policy_fc.bias"), dim=-1)
    vh = F.relu(bn(F.conv2d(h, t("value_head.value_head.weight")), sd, "value_head.value_bn")).flatten(1)
    v1 = F.relu(vh @ t("value_head.value_fc1.weight").T + t("value_head.value_fc1.bias"))
    val = torch.tanh(v1 @ t("value_head.value_fc2.weight").T + t("value_head.value_fc2.bias")).squeeze(-1)
    return pol, val


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    print("%-28s %-9s %12s %12s" % ("network", "variant", "max |dp|", "max |dv|"))
    for nb, B, seed in ((4, 9, 77), (4, 9, 3), (10, 9, 5), (10, 15, 8)):
        sd = pvnet_weights.make_state_dict(nb, 5, 128, B, seed)
        x = torch.from_numpy((rs.rand(24, 5, B, B) < 0.3).astype(np.float64))
        p0, v0 = forward(sd, x, nb, "exact")
        for mode in ("split3", "fp8corr", "hi_only"):
            p, v = forward(sd, x, nb, mode)
            print("%-28s %-9s %12.2e %12.2e" % ("%d blocks, %dx%d, seed %d" % (nb, B, B, seed), mode,
                                                (p - p0).abs().max().item(), (v - v0).abs().max().item()))
