"""Checkpoint importer against a synthetic state dict with the reference's key layout (no real checkpoint ships:
/root/reference/.MISSING_LARGE_BLOBS; the reference itself falls back to random weights, homomorphic_eval.py:254-256)."""
import numpy as np
import torch


def _fake_state(model, rng):
    st = {}
    def conv(name, layer): st[f"module.feature.trunk.{name}.weight"] = torch.from_numpy(rng.normal(0, .1, layer.weight.shape)).float()
    def bn(name, c):
        for f, v in (("weight", rng.uniform(.5, 1.5, c)), ("bias", rng.normal(0, .1, c)), ("running_mean", rng.normal(0, 1, c)), ("running_var", rng.uniform(.5, 2, c))):
            st[f"module.feature.trunk.{name}.{f}"] = torch.from_numpy(v).float()
        st[f"module.feature.trunk.{name}.num_batches_tracked"] = torch.tensor(7)
    conv("1", model.conv1); bn("2", model.conv1.weight.shape[0])
    st["module.feature.trunk.0.act_quant.fused_activation_quant_proxy.tensor_quant.scaling_impl.value"] = torch.tensor(0.1)   # Brevitas scale: ignored
    for i, b in enumerate(model.blocks):
        n = 5 + i
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
    assert all("act_quant" in k or "num_batches_tracked" in k for k in unused)
    # and the imported model compiles with the stored BatchNorm statistics (they are not re-calibrated)
    c = cc.compile_model(model, synthetic_dct_batch(16, seed=7))
    assert c.max_bit_width <= 16 and np.allclose(model.bn1.mean, state["module.feature.trunk.2.running_mean"].numpy())
