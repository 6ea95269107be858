"""torch.nn.Module trunk -> dctfhe.models.ResNetQ: what lets the reference hand over ITS model object.

The reference compiles `model.module.feature` (homomorphic_eval.py:276-277 QAT, :287-288 PTQ): a `ResNetQDCT` /
`ResNetDCT` (models/backbone.py:187-288 / :107-184) whose `.trunk` is an `nn.Sequential` of
    [QuantIdentity] Conv2d BatchNorm2d [ReLU] [QuantIdentity] block* AvgPool2d [QuantIdentity] Flatten
with residual blocks that expose `C1, BN1, C2, BN2` and, when the channel count changes, `shortcut, BNshortcut`
(`SimpleBlock` :18-58, `SimpleQBlock` :61-104).  The walk below is duck-typed -- it looks at class names and attributes,
never at the classes themselves -- so the pure-torch float models import unchanged, and so do Brevitas modules where
that package exists (QuantConv2d is a Conv2d, QuantReLU/QuantIdentity are recognised by name).  Quantiser scales a
Brevitas module carries are picked up by dctfhe.checkpoint from the state dict; here only weights, BatchNorm affine +
running statistics and the layer geometry are read.
"""
import numpy as np

from . import models


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _cls(m):
    return type(m).__name__


def _conv(m):
    if getattr(m, "bias", None) is not None:
        raise ValueError("convolutions with a bias are not part of the encrypted trunk (reference backbone.py:220 `bias: False`)")
    k, s, p = m.kernel_size, m.stride, m.padding
    if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or tuple(getattr(m, "dilation", (1, 1))) != (1, 1) or getattr(m, "groups", 1) != 1:
        raise ValueError(f"unsupported convolution geometry: kernel {k} stride {s} padding {p}")
    return models.ConvLayer(_np(m.weight), int(s[0]), int(p[0]))


def _bn(m):
    if m.running_mean is None:
        raise ValueError("BatchNorm2d without running statistics")
    g = _np(m.weight) if m.weight is not None else np.ones(m.num_features)
    b = _np(m.bias) if m.bias is not None else np.zeros(m.num_features)
    return models.BatchNorm(gamma=g, beta=b, mean=_np(m.running_mean), var=_np(m.running_var), eps=float(m.eps))


def _is_block(m):
    return all(hasattr(m, a) for a in ("C1", "BN1", "C2", "BN2"))


def _block(m):
    b = models.QBlock(C1=_conv(m.C1), BN1=_bn(m.BN1), C2=_conv(m.C2), BN2=_bn(m.BN2))
    if getattr(m, "shortcut_type", "identity" if not hasattr(m, "shortcut") else "1x1") != "identity":
        b.shortcut, b.BNshortcut = _conv(m.shortcut), _bn(m.BNshortcut)
    if b.C2.stride != 1 or b.C1.pad != 1 or b.C2.pad != 1 or b.C1.weight.shape[2] != 3:
        raise ValueError("residual block is not the 3x3/3x3 SimpleBlock shape")
    return b


def is_torch_module(obj):
    try:
        import torch.nn as nn
    except Exception:
        return False
    return isinstance(obj, nn.Module)


def from_torch_module(module, bit_width=4, in_channels=None, img_size=None, name=None, classifier=None):
    """module: the trunk (`model.module.feature`), its `.trunk` Sequential, or any module whose children are the layers in
    order.  classifier: optional nn.Linear (reference utils.py:22) whose weights ride along for the clear classification.
    img_size is only recorded (the calibration batch fixes the real size)."""
    seq = getattr(module, "trunk", module)
    qargs = getattr(module, "qconv_args", None)          # reference ResNetQDCT keeps its Brevitas arguments (backbone.py:217-223)
    if isinstance(qargs, dict) and "weight_bit_width" in qargs:
        bit_width = int(qargs["weight_bit_width"])
    layers = list(seq.children())
    conv1 = bn1 = None
    relu1 = False
    blocks, avgpool = [], None
    for m in layers:
        c = _cls(m)
        if _is_block(m):
            blocks.append(_block(m))
        elif "Conv" in c and hasattr(m, "weight"):
            if conv1 is not None or blocks:
                raise ValueError("only the stem convolution may stand outside a residual block")
            conv1 = _conv(m)
        elif "BatchNorm" in c:
            if bn1 is not None or blocks:
                raise ValueError("only the stem BatchNorm may stand outside a residual block")
            bn1 = _bn(m)
        elif "ReLU" in c:
            if blocks:
                raise ValueError("stand-alone ReLU after the residual blocks")
            relu1 = True
        elif "AvgPool" in c:
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            st = m.stride if isinstance(m.stride, int) else m.stride[0]
            if st != k or (m.padding if isinstance(m.padding, int) else m.padding[0]) != 0:
                raise ValueError("AvgPool2d must be non-overlapping and unpadded (reference backbone.py:276)")
            avgpool = int(k)
        elif "MaxPool" in c:
            raise ValueError("MaxPool2d in the trunk (pool1_kernel perturbations, reference backbone.py:252-258) has no encrypted operator here")
        elif "QuantIdentity" in c or "Flatten" in c or "Identity" in c or "Dropout" in c:
            continue            # quantisers are re-stated by the compiler; Flatten is a view
        else:
            raise ValueError(f"unsupported layer in the trunk: {c}")
    if conv1 is None or bn1 is None or avgpool is None:
        raise ValueError("trunk must contain the stem Conv2d + BatchNorm2d and a final AvgPool2d")
    cin = conv1.weight.shape[1]
    if in_channels is not None and in_channels != cin:
        raise ValueError(f"in_channels {in_channels} but the stem takes {cin}")
    feat = (blocks[-1].C2.weight.shape[0] if blocks else conv1.weight.shape[0])
    if img_size is not None:
        s = (img_size + 2 * conv1.pad - conv1.weight.shape[2]) // conv1.stride + 1
        for b in blocks:
            s = (s + 2 - 3) // b.C1.stride + 1
        feat *= (s // avgpool) ** 2
    cw = cb = None
    if classifier is not None:
        cw = _np(classifier.weight)
        cb = _np(classifier.bias) if classifier.bias is not None else np.zeros(cw.shape[0])
    return models.ResNetQ(name=name or _cls(module), in_channels=cin, img_size=img_size or 0, bit_width=bit_width, conv1=conv1, bn1=bn1,
                          relu1=relu1, blocks=blocks, avgpool_kernel=avgpool, final_feat_dim=feat, classifier_w=cw, classifier_b=cb)


def seed_parameters(module, seed):
    """Deterministic, generator-independent fill of every parameter and BatchNorm statistic, in state_dict order (tests and
    tools/make_goldens.py: the same numbers in the reference's module and in a twin without sharing torch's RNG stream)."""
    import torch
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            if k.endswith("running_var") or (k.endswith(".weight") and v.dim() == 1):
                a = rng.uniform(0.5, 1.5, tuple(v.shape))
            elif v.dim() == 4:
                a = rng.normal(0.0, np.sqrt(2.0 / (v.shape[0] * v.shape[2] * v.shape[3])), tuple(v.shape))
            else:
                a = rng.normal(0.0, 0.1, tuple(v.shape))
            v.copy_(torch.from_numpy(a).to(v.dtype))
    return module
