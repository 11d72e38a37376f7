"""tests/golden/trained_2block_9x9.npz from a checkpoint of tools/train_omok.py (2 blocks, 128 planes, 9x9; run A of round 4:
`tools/exp/r4a.sh`, 111 iterations x 1024 self-play games at 200 sims on the MI355X, 64 : 0 against its iteration 0).

The fixture is DATA: the state_dict's float tensors rounded to bfloat16 and stored as their 16-bit patterns (half the
bytes; a network whose weights are bf16-representable is still the sharp, trained policy the GPU test needs -- the test
evaluates exactly these weights on both sides), integer buffers as they are.

    python tools/make_trained_fixture.py gpurun_out/r4a_train/final.pt tests/golden/trained_2block_9x9.npz
"""
import sys

import numpy as np
import torch


def main(src, dst):
    sd = torch.load(src, map_location="cpu")
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            bits = v.float().to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)
            out["bf16:" + k] = bits
        else:
            out["int:" + k] = v.numpy()
    np.savez_compressed(dst, **out)
    print(dst, sum(a.size for a in out.values()), "values")


def load(path):
    """name -> float32 / int64 numpy arrays (the reference's state_dict keys)."""
    z = np.load(path)
    sd = {}
    for k in z.files:
        kind, name = k.split(":", 1)
        a = z[k]
        sd[name] = (a.astype(np.uint32) << 16).view(np.float32) if kind == "bf16" else a
    return sd


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
