"""Circuit topology against the reference's float twin (tests/golden/topology_golden.npz): every conv's
channels / kernel / stride / padding / input and output size, the trunk output size and the trunk's
conv+BN parameter count."""
import os

import numpy as np
import pytest

T = np.load(os.path.join(os.path.dirname(__file__), "golden", "topology_golden.npz"))


def _convs(model):
    from dctfhe import compile as cc
    rows = []
    s = model.img_size
    def one(layer, s):
        co, ci, k, _ = layer.weight.shape
        so = (s + 2 * layer.pad - k) // layer.stride + 1
        rows.append((ci, co, k, layer.stride, layer.pad, s, s, so, so))
        return so
    s = one(model.conv1, s)
    for b in model.blocks:
        s1 = one(b.C1, s)
        one(b.C2, s1)
        if b.shortcut is not None:
            one(b.shortcut, s)
        s = s1
    return np.array(rows, np.int64), s


@pytest.mark.parametrize("tag,fn,cin,size", [("r20_24_16", "ResNet20QAT", 24, 16), ("r20_3_32", "ResNet20QAT", 3, 32),
                                             ("r18_3_32", "ResNet18QAT", 3, 32), ("r18_48_112", "ResNet18QAT", 48, 112)])
def test_conv_topology(tag, fn, cin, size):
    from dctfhe import models
    m = getattr(models, fn)(bit_width=4, in_channels=cin, img_size=size)
    rows, s = _convs(m)
    assert np.array_equal(rows, T[f"{tag}_convs"])
    feat = m.blocks[-1].C2.weight.shape[0] * (s // m.avgpool_kernel) ** 2
    assert feat == int(np.prod(T[f"{tag}_out"][1:])) == m.final_feat_dim
    nparam = m.conv1.weight.size + 2 * m.bn1.gamma.size
    for b in m.blocks:
        nparam += b.C1.weight.size + b.C2.weight.size + 2 * b.BN1.gamma.size + 2 * b.BN2.gamma.size
        if b.shortcut is not None:
            nparam += b.shortcut.weight.size + 2 * b.BNshortcut.gamma.size
    assert nparam == int(T[f"{tag}_params"])


def test_r20_counts_match_survey():
    from dctfhe import models
    rows, _ = _convs(models.ResNet20QAT(4, 24, 16))
    macs = int((rows[:, 0] * rows[:, 1] * rows[:, 2] ** 2 * rows[:, 7] * rows[:, 8]).sum())
    outs = int((rows[:, 1] * rows[:, 7] * rows[:, 8]).sum())
    assert len(rows) == 21 and outs == 215040 and abs(macs - 89.2e6) < 0.1e6
