"""Importer for the reference's training checkpoints.

The reference saves `{'epoch', 'state', 'prec1', 'prec5', 'optimizer'}` with torch.save (train.py:82-89,118-126) and
the evaluator reads `checkpoint['state']` into `nn.DataParallel(BaselineTrain(...))` (homomorphic_eval.py:247-253), so
keys look like `module.feature.trunk.<i>.<...>` and `module.classifier.{weight,bias}`.  Trunk indices follow
ResNetQDCT.__init__ (backbone.py:229-281): 0 quant_inp, 1 conv1, 2 bn1, [3 relu], quant_out, then one SimpleQBlock per
entry (attributes C1, BN1, C2, BN2, shortcut, BNshortcut; backbone.py:61-91), avgpool, QuantIdentity, Flatten.

What is imported: convolution weights, BatchNorm affine + running statistics, the clear classifier, and the learned
activation-quantiser scales: every `...act_quant.fused_activation_quant_proxy.tensor_quant.scaling_impl.value` (the
learned threshold of `Int8ActPerTensorFloat` / `QuantReLU`, backbone.py:224-227) becomes the integer step
    scale = |value| / 2^(bits-1)        signed quantisers (QuantIdentity)
    scale = |value| / (2^bits - 1)      unsigned quantisers (QuantReLU)
[K: Brevitas 0.8 `IntScaling` -- -min_int for signed, max_int for unsigned -- with the float (non power-of-two) scaling
restriction; unverifiable here, Brevitas is absent] and lands in `model.act_scales`, which dctfhe.compile prefers over its
calibration fallback.  Sites whose key is missing are re-calibrated from the calibration batch (the documented fallback).
Weight scales follow from the weights (stats-based in the reference too).  Files are opened with
torch.load(weights_only=True) only.
"""
import re

import numpy as np
import torch

from . import models


def load_state(path):
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    state = ckpt["state"] if isinstance(ckpt, dict) and "state" in ckpt else ckpt
    meta = {k: ckpt[k] for k in ("epoch", "prec1", "prec5") if isinstance(ckpt, dict) and k in ckpt}
    return state, meta


def _strip(key):
    for pre in ("module.", "feature.", "trunk."):
        if key.startswith(pre):
            key = key[len(pre):]
    return key


def import_into(model, state):
    """Overwrite the weights of a dctfhe.models.ResNetQ with those of a reference state dict; returns the unused keys."""
    trunk, cls = {}, {}
    for k, v in state.items():
        kk = k[len("module."):] if k.startswith("module.") else k
        if kk.startswith("classifier."):
            cls[kk[len("classifier."):]] = v
        elif kk.startswith("feature."):
            trunk[_strip(kk)] = v
    used = set()

    def take(name):
        used.add(name)
        return trunk[name].detach().cpu().numpy().astype(np.float64)

    def load_bn(bn, prefix):
        bn.gamma, bn.beta = take(prefix + ".weight"), take(prefix + ".bias")
        bn.mean, bn.var = take(prefix + ".running_mean"), take(prefix + ".running_var")

    idx = sorted({int(m.group(1)) for k in trunk for m in [re.match(r"(\d+)\.", k)] if m})
    block_idx = [i for i in idx if any(k.startswith(f"{i}.C1.") for k in trunk)]
    if len(block_idx) != len(model.blocks):
        raise ValueError(f"checkpoint has {len(block_idx)} residual blocks, the model {len(model.blocks)}")
    model.conv1.weight = take("1.weight")
    load_bn(model.bn1, "2")
    for bi, blk in zip(block_idx, model.blocks):
        blk.C1.weight, blk.C2.weight = take(f"{bi}.C1.weight"), take(f"{bi}.C2.weight")
        load_bn(blk.BN1, f"{bi}.BN1")
        load_bn(blk.BN2, f"{bi}.BN2")
        if blk.shortcut is not None:
            blk.shortcut.weight = take(f"{bi}.shortcut.weight")
            load_bn(blk.BNshortcut, f"{bi}.BNshortcut")
    # learned activation scales: trunk index [+ block attribute] -> the compiler's site names
    bits = model.bit_width
    pre_block = [i for i in idx if i < (block_idx[0] if block_idx else 10 ** 9)]
    act_keys = {}
    for k in trunk:
        m = re.match(r"(\d+)\.(?:(\w+)\.)?act_quant\..*scaling_impl\.value$", k)
        if m:
            act_keys[(int(m.group(1)), m.group(2))] = k
    plain = sorted(i for (i, sub) in act_keys if sub is None)
    names = {}
    before = [i for i in plain if i in pre_block or not block_idx or i < block_idx[0]]
    after = [i for i in plain if block_idx and i > block_idx[-1]]
    if before:
        names[before[0]] = ("quant_inp", True)                  # trunk[0] QuantIdentity (backbone.py:231)
        rest = before[1:]
        if model.relu1 and len(rest) >= 2:
            names[rest[0]] = ("stem_relu", False)               # QuantReLU (backbone.py:249)
            names[rest[1]] = ("stem_quant_out", True)           # quant_out (backbone.py:258-261)
        elif rest:
            names[rest[-1]] = ("stem_quant_out", True)
    if after:
        names[after[-1]] = ("final", True)                      # QuantIdentity after the pooling (backbone.py:278)
    for (i, sub), k in act_keys.items():
        if sub is None:
            if i not in names:
                continue
            site, signed = names[i]
        else:
            if i not in block_idx or sub not in ("relu1", "relu2", "quant_out", "BNquant_out"):
                continue
            site, signed = ("block", block_idx.index(i), sub), sub in ("quant_out", "BNquant_out")
        v = abs(float(trunk[k].detach().cpu().reshape(-1)[0]))
        used.add(k)
        if v > 0:
            model.act_scales[site] = v / (2 ** (bits - 1) if signed else 2 ** bits - 1)
    if "weight" in cls:
        model.classifier_w = cls["weight"].detach().cpu().numpy().astype(np.float64)
        model.classifier_b = cls["bias"].detach().cpu().numpy().astype(np.float64) if "bias" in cls else np.zeros(model.classifier_w.shape[0])
    return sorted(set(trunk) - used)


def load_checkpoint(path, model):
    state, meta = load_state(path)
    unused = import_into(model, state)
    return meta, unused
