"""Deterministic PVNet weights for tests (numpy RandomState; no torch init stream involved).

Key names and shapes are those of the reference's ``model.PVNet.state_dict()``
(/root/reference/2_AlphaOmok/model.py:76-104). Used by
tools/gen_golden.py (to load the reference net) and by the tests (to load ours), so both sides
see bit-identical parameters, including non-trivial BatchNorm running statistics.
"""
import numpy as np


def make_state_dict(n_block, inplanes, planes, board_size, seed):
    rng = np.random.RandomState(seed)
    A = board_size * board_size
    sd = {}

    def conv(name, cout, cin, k):
        fan_in = cin * k * k
        sd[name] = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32)

    def bn(prefix, c):
        sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[prefix + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[prefix + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
        sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[prefix + ".num_batches_tracked"] = np.array(7, np.int64)

    def fc(prefix, cout, cin):
        sd[prefix + ".weight"] = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
        sd[prefix + ".bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)

    conv("conv1.weight", planes, inplanes, 3)
    bn("bn1", planes)
    for i in range(n_block):
        conv("layers.%d.conv1.weight" % i, planes, planes, 3)
        bn("layers.%d.bn1" % i, planes)
        conv("layers.%d.conv2.weight" % i, planes, planes, 3)
        bn("layers.%d.bn2" % i, planes)
    conv("policy_head.policy_head.weight", 2, planes, 1)
    bn("policy_head.policy_bn", 2)
    fc("policy_head.policy_fc", A, 2 * A)
    conv("value_head.value_head.weight", 1, planes, 1)
    bn("value_head.value_bn", 1)
    fc("value_head.value_fc1", planes, A)
    fc("value_head.value_fc2", 1, planes)
    return sd
