"""Checkpoint importer against a synthetic state dict with the reference's key layout (no real checkpoint ships:
/root/reference/.MISSING_LARGE_BLOBS; the reference itself falls back to random weights, homomorphic_eval.py:254-256)."""
import numpy as np
import torch


def _fake_state(model, rng, act_scales=True):
    st = {}
    def conv(name, layer): st[f"module.feature.trunk.{name}.weight"] = torch.from_numpy(rng.normal(0, .1, layer.weight.shape)).float()
    def bn(name, c):
        for f, v in (("weight", rng.uniform(.5, 1.5, c)), ("bias", rng.normal(0, .1, c)), ("running_mean", rng.normal(0, 1, c)), ("running_var", rng.uniform(.5, 2, c))):
            st[f"module.feature.trunk.{name}.{f}"] = torch.from_numpy(v).float()
        st[f"module.feature.trunk.{name}.num_batches_tracked"] = torch.tensor(7)
    conv("1", model.conv1); bn("2", model.conv1.weight.shape[0])
    AQ = "act_quant.fused_activation_quant_proxy.tensor_quant.scaling_impl.value"
    if act_scales:        # Brevitas learned thresholds, laid out as ResNetQDCT.__init__ does (backbone.py:229-281)
        st[f"module.feature.trunk.0.{AQ}"] = torch.tensor(2.0)            # quant_inp
        st[f"module.feature.trunk.3.{AQ}"] = torch.tensor(3.0)            # QuantReLU
        st[f"module.feature.trunk.4.{AQ}"] = torch.tensor([1.6])          # quant_out
        st[f"module.feature.trunk.{5 + len(model.blocks) + 1}.{AQ}"] = torch.tensor(0.8)     # QuantIdentity after avgpool
    for i, b in enumerate(model.blocks):
        n = 5 + i
        if act_scales:
            st[f"module.feature.trunk.{n}.relu1.{AQ}"] = torch.tensor(2.25)
            st[f"module.feature.trunk.{n}.relu2.{AQ}"] = torch.tensor(3.75)
            st[f"module.feature.trunk.{n}.quant_out.{AQ}"] = torch.tensor(1.2)
            if b.shortcut is not None:
                st[f"module.feature.trunk.{n}.BNquant_out.{AQ}"] = torch.tensor(-1.0)   # the threshold enters by magnitude
        conv(f"{n}.C1", b.C1); conv(f"{n}.C2", b.C2); bn(f"{n}.BN1", b.C1.weight.shape[0]); bn(f"{n}.BN2", b.C2.weight.shape[0])
        if b.shortcut is not None:
            conv(f"{n}.shortcut", b.shortcut); bn(f"{n}.BNshortcut", b.shortcut.weight.shape[0])
    st["module.classifier.weight"] = torch.from_numpy(rng.normal(0, 1, model.classifier_w.shape)).float()
    st["module.classifier.bias"] = torch.zeros(model.classifier_w.shape[0])
    return st


def test_import_reference_layout(tmp_path):
    from dctfhe import checkpoint, compile as cc, models
    from dctfhe.synthetic import synthetic_dct_batch
    rng = np.random.default_rng(0)
    model = models.ResNet20QAT(4, 24, 16)
    state = _fake_state(model, rng)
    path = str(tmp_path / "best.tar")
    torch.save({"epoch": 3, "state": state, "prec1": 90.5, "prec5": 99.0, "optimizer": {}}, path)
    meta, unused = checkpoint.load_checkpoint(path, model)
    assert meta["epoch"] == 3 and abs(meta["prec1"] - 90.5) < 1e-6
    assert np.allclose(model.conv1.weight, state["module.feature.trunk.1.weight"].numpy())
    assert np.allclose(model.blocks[3].BNshortcut.mean, state["module.feature.trunk.8.BNshortcut.running_mean"].numpy())
    assert np.allclose(model.classifier_w, state["module.classifier.weight"].numpy())
    assert all("num_batches_tracked" in k for k in unused), unused
    # learned activation scales: |threshold| / 8 for the signed 4-bit quantisers, / 15 for the QuantReLUs
    a = model.act_scales
    near = lambda x, y: abs(x - y) < 1e-7            # the thresholds are stored as float32
    assert near(a["quant_inp"], 2.0 / 8) and near(a["stem_relu"], 3.0 / 15) and near(a["stem_quant_out"], 1.6 / 8) and near(a["final"], 0.8 / 8)
    assert near(a[("block", 0, "relu1")], 2.25 / 15) and near(a[("block", 8, "relu2")], 3.75 / 15) and near(a[("block", 4, "quant_out")], 1.2 / 8)
    assert near(a[("block", 3, "BNquant_out")], 1.0 / 8) and ("block", 0, "BNquant_out") not in a
    # and the imported model compiles with the stored BatchNorm statistics (they are not re-calibrated)
    c = cc.compile_model(model, synthetic_dct_batch(16, seed=7))
    assert c.max_bit_width <= 16 and np.allclose(model.bn1.mean, state["module.feature.trunk.2.running_mean"].numpy())
    # ... and with the imported scales, not re-calibrated ones
    assert abs(c.in_scale - 2.0 / 8) < 1e-7 and abs(c.out_scale - 0.8 / 8) < 1e-7


def test_missing_scales_fall_back_to_calibration(tmp_path):
    from dctfhe import checkpoint, compile as cc, models
    from dctfhe.synthetic import synthetic_dct_batch
    model = models.ResNet20QAT(4, 24, 16)
    path = str(tmp_path / "best.tar")
    torch.save({"state": _fake_state(model, np.random.default_rng(0), act_scales=False)}, path)
    checkpoint.load_checkpoint(path, model)
    assert model.act_scales == {}
    calib = synthetic_dct_batch(16, seed=7)
    c = cc.compile_model(model, calib)
    assert abs(c.in_scale - float(np.abs(calib).max()) / 7) < 1e-9
