"""The boundary takes the reference's model OBJECT: `compile_torch_model(model.module.feature, ...)` /
`compile_brevitas_qat_model(...)` (reference homomorphic_eval.py:276-295) are handed a torch.nn.Module trunk.

tests/golden/torch_import_golden.npz (tools/make_goldens.py, build container) holds, for the reference's float twin
`ResNet20(24,16)` / `ResNet18(3,32)` (models/backbone.py:291-327) under seeded weights: the state-dict key names and shapes,
a seeded input and the reference's float forward output.  Here a twin written for this test (same attribute names, nothing
copied) must carry the same state-dict naming, and `from_torch_module(twin)` evaluated in float must reproduce the reference
output: stem, block order, shortcut type, stride placement, BatchNorm statistics, floor-mode pooling.
Where /root/reference exists (build container only) the reference's own classes are walked unchanged as well."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_import_golden.npz"))
REF = "/root/reference/dct-cryptonets"


class TwinBlock(nn.Module):
    def __init__(self, cin, cout, half):
        super().__init__()
        self.C1 = nn.Conv2d(cin, cout, 3, 2 if half else 1, 1, bias=False)
        self.BN1 = nn.BatchNorm2d(cout)
        self.relu1 = nn.ReLU()
        self.C2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.BN2 = nn.BatchNorm2d(cout)
        self.relu2 = nn.ReLU()
        self.shortcut_type = "identity"
        if cin != cout:
            self.shortcut = nn.Conv2d(cin, cout, 1, 2 if half else 1, bias=False)
            self.BNshortcut = nn.BatchNorm2d(cout)
            self.shortcut_type = "1x1"


class TwinTrunk(nn.Module):
    def __init__(self, layers, dims, cin, stem, avg, skip_single):
        super().__init__()
        k, s, p = stem
        trunk = [nn.Conv2d(cin, dims[0], k, s, p, bias=False), nn.BatchNorm2d(dims[0]), nn.ReLU()]
        c = dims[0]
        for i, n in enumerate(layers):
            for j in range(n):
                trunk.append(TwinBlock(c, dims[i], (i >= (2 if skip_single else 1)) and j == 0))
                c = dims[i]
        trunk += [nn.AvgPool2d(avg), nn.Flatten()]
        self.trunk = nn.Sequential(*trunk)


TWINS = {"r20_24_16": lambda: TwinTrunk([3, 3, 3], [48, 56, 64], 24, (1, 1, 0), 7, True),
         "r18_3_32": lambda: TwinTrunk([2, 2, 2, 2], [64, 128, 256, 512], 3, (3, 1, 1), 3, False)}
SIZES = {"r20_24_16": 16, "r18_3_32": 32}


@pytest.mark.parametrize("tag", sorted(TWINS))
def test_twin_module_import_reproduces_reference_forward(tag):
    from dctfhe import models
    from dctfhe.torch_import import from_torch_module, seed_parameters
    twin = seed_parameters(TWINS[tag](), 11).eval()
    sd = twin.state_dict()
    assert list(sd.keys()) == G[f"{tag}_keys"].tolist()                       # the reference's naming: trunk.<i>.C1.weight, ...
    assert [";".join(map(str, v.shape)) for v in sd.values()] == G[f"{tag}_shapes"].tolist()
    m = from_torch_module(twin, bit_width=4, img_size=SIZES[tag])
    ours = getattr(models, "ResNet20QAT" if tag.startswith("r20") else "ResNet18QAT")(4, m.in_channels, SIZES[tag])
    assert len(m.blocks) == len(ours.blocks) and m.avgpool_kernel == ours.avgpool_kernel and m.relu1 == ours.relu1
    assert m.final_feat_dim == ours.final_feat_dim == G[f"{tag}_y"].shape[1]
    for a, b in zip(m.blocks, ours.blocks):
        assert (a.shortcut is None) == (b.shortcut is None) and a.C1.stride == b.C1.stride and a.C1.weight.shape == b.C1.weight.shape
    y = models.float_forward(m, G[f"{tag}_x"])
    assert np.allclose(y, G[f"{tag}_y"], rtol=2e-4, atol=2e-4), np.abs(y - G[f"{tag}_y"]).max()


def test_compile_entry_points_take_the_module():
    """compile_torch_model / compile_brevitas_qat_model given the nn.Module == given the imported description (same blob)"""
    from dctfhe import compile as cc
    from dctfhe.quantized_module import compile_brevitas_qat_model, compile_torch_model
    from dctfhe.synthetic import synthetic_dct_batch
    from dctfhe.torch_import import from_torch_module, seed_parameters
    twin = seed_parameters(TWINS["r20_24_16"](), 11).eval()
    calib = synthetic_dct_batch(24, seed=7)
    want = cc.compile_model(from_torch_module(twin, bit_width=4), calib, rounding_threshold_bits=6, n_bits=5, p_error=0.01)
    qat = compile_brevitas_qat_model(twin, torch.from_numpy(calib), n_bits=5, rounding_threshold_bits=6, p_error=0.01)
    assert qat.compiled.blob == want.blob
    ptq = compile_torch_model(twin, torch.from_numpy(calib), n_bits=5, rounding_threshold_bits=6, p_error=0.01, bit_width=4)
    assert ptq.compiled.blob == want.blob
    assert qat.fhe_circuit.graph.maximum_integer_bit_width() <= 16 and "round_lut" in qat.fhe_circuit.mlir


def test_unsupported_trunks_are_refused():
    from dctfhe.torch_import import from_torch_module
    with pytest.raises(ValueError, match="MaxPool2d"):
        from_torch_module(nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.MaxPool2d(3, 2, 1), nn.AvgPool2d(2)))
    with pytest.raises(ValueError, match="bias"):
        from_torch_module(nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.AvgPool2d(2)))
    with pytest.raises(ValueError, match="unsupported layer"):
        from_torch_module(nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Sigmoid(), nn.AvgPool2d(2)))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_reference_classes_import_unchanged():
    """the reference's own `ResNet20(24,16)` object (models/backbone.py:291-302) through the boundary, bit-identical to the twin"""
    from unittest.mock import MagicMock
    from dctfhe.quantized_module import compile_torch_model
    from dctfhe.synthetic import synthetic_dct_batch
    from dctfhe.torch_import import from_torch_module, seed_parameters
    sys.dont_write_bytecode = True
    added = []
    for name in ("brevitas", "brevitas.nn", "brevitas.quant"):
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
            added.append(name)
    sys.path.insert(0, REF)
    try:
        bb = importlib.import_module("models.backbone")
        ref = seed_parameters(bb.ResNet20(in_channels=24, img_size=16), 11).eval()
        twin = seed_parameters(TWINS["r20_24_16"](), 11).eval()
        a, b = from_torch_module(ref), from_torch_module(twin)
        assert np.array_equal(a.conv1.weight, b.conv1.weight) and a.relu1 == b.relu1 and a.avgpool_kernel == b.avgpool_kernel
        for x, y in zip(a.blocks, b.blocks):
            assert np.array_equal(x.C2.weight, y.C2.weight) and np.array_equal(x.BN2.var, y.BN2.var) and (x.shortcut is None) == (y.shortcut is None)
        calib = synthetic_dct_batch(16, seed=7)
        qa = compile_torch_model(ref, torch.from_numpy(calib), n_bits=5, rounding_threshold_bits=6, p_error=0.01, bit_width=4)
        qb = compile_torch_model(twin, torch.from_numpy(calib), n_bits=5, rounding_threshold_bits=6, p_error=0.01, bit_width=4)
        assert qa.compiled.blob == qb.compiled.blob
    finally:
        sys.path.remove(REF)
        for name in added:
            sys.modules.pop(name, None)
        for name in [n for n in sys.modules if n == "models" or n.startswith("models.")]:
            sys.modules.pop(name, None)
