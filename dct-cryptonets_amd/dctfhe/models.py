"""Model descriptions for the encrypted trunk.

Restates the topology of the reference's quantised ResNets (reference models/backbone.py:187-288
`ResNetQDCT`, :61-104 `SimpleQBlock`, :305-342 factories, :347-582 `all_network_perturbations`) as
plain data: float weights + BatchNorm statistics per layer.  Brevitas (reference env.yml:31) is not
available here, so the quantisers it would attach are restated in dctfhe/compile.py.  Only the
entries of `all_network_perturbations` the BASELINE configs need are listed.
"""
import math
from dataclasses import dataclass, field

import numpy as np

# reference models/backbone.py:376-383, 393-401, 439-446, 555-563
NET_PERTURBATIONS = {
    "48_3_32": dict(conv1_kernel=3, conv1_stride=1, conv1_padding=1, relu1=True, avgpool_kernel=7),
    "48_24_16": dict(conv1_kernel=1, conv1_stride=1, conv1_padding=0, relu1=True, avgpool_kernel=7),
    "48_24_8": dict(conv1_kernel=1, conv1_stride=1, conv1_padding=0, relu1=True, avgpool_kernel=3),
    "64_3_32": dict(conv1_kernel=3, conv1_stride=1, conv1_padding=1, relu1=True, avgpool_kernel=3),
    "64_48_112": dict(conv1_kernel=1, conv1_stride=1, conv1_padding=0, relu1=False, avgpool_kernel=14),
}


@dataclass
class ConvLayer:
    weight: np.ndarray            # float64 [Cout, Cin, K, K]
    stride: int
    pad: int


@dataclass
class BatchNorm:
    gamma: np.ndarray
    beta: np.ndarray
    mean: np.ndarray = None       # running stats; None -> taken from the calibration pass
    var: np.ndarray = None
    eps: float = 1e-5


@dataclass
class QBlock:                     # reference SimpleQBlock, backbone.py:61-104
    C1: ConvLayer
    BN1: BatchNorm
    C2: ConvLayer
    BN2: BatchNorm
    shortcut: ConvLayer = None    # 1x1 conv when indim != outdim (backbone.py:80-83)
    BNshortcut: BatchNorm = None


@dataclass
class ResNetQ:                    # reference ResNetQDCT, backbone.py:187-288
    name: str
    in_channels: int
    img_size: int
    bit_width: int
    conv1: ConvLayer
    bn1: BatchNorm
    relu1: bool
    blocks: list
    avgpool_kernel: int
    final_feat_dim: int
    classifier_w: np.ndarray = None   # clear nn.Linear (reference utils.py:22), [classes, feat]
    classifier_b: np.ndarray = None
    # learned activation-quantiser scales (real value of one integer step), imported from a Brevitas checkpoint by
    # dctfhe.checkpoint; keys: "quant_inp", "stem_relu", "stem_quant_out", ("block", i, "relu1" | "quant_out" |
    # "BNquant_out" | "relu2"), "final".  A missing key falls back to calibration (dctfhe.compile.act_scale).
    act_scales: dict = field(default_factory=dict)


def _init_conv(rng, cout, cin, k):
    # reference init_layer (backbone.py:8-15): N(0, sqrt(2 / (k*k*cout)))
    return rng.normal(0.0, math.sqrt(2.0 / float(k * k * cout)), size=(cout, cin, k, k))


def _init_bn(c):
    return BatchNorm(gamma=np.ones(c), beta=np.zeros(c))


def build_resnet_q(list_num_layers, list_out_dims, in_channels, img_size, bit_width=4, skip_single_downsample=False,
                   num_classes=10, seed=0, name="ResNetQ"):
    """Random-weight model with the reference's initialisation (checkpoints are not shipped: the
    reference itself falls back to random weights, homomorphic_eval.py:254-256)."""
    key = f"{list_out_dims[0]}_{in_channels}_{img_size}"
    if key not in NET_PERTURBATIONS:
        raise KeyError(f"no network perturbation entry for {key}")
    pert = NET_PERTURBATIONS[key]
    rng = np.random.default_rng(seed)
    conv1 = ConvLayer(_init_conv(rng, list_out_dims[0], in_channels, pert["conv1_kernel"]), pert["conv1_stride"], pert["conv1_padding"])
    blocks = []
    indim = list_out_dims[0]
    for i, nl in enumerate(list_num_layers):
        for j in range(nl):
            half_res = ((i >= 2) if skip_single_downsample else (i >= 1)) and j == 0   # backbone.py:264-270
            outdim = list_out_dims[i]
            b = QBlock(C1=ConvLayer(_init_conv(rng, outdim, indim, 3), 2 if half_res else 1, 1), BN1=_init_bn(outdim),
                       C2=ConvLayer(_init_conv(rng, outdim, outdim, 3), 1, 1), BN2=_init_bn(outdim))
            if indim != outdim:
                b.shortcut = ConvLayer(_init_conv(rng, outdim, indim, 1), 2 if half_res else 1, 0)
                b.BNshortcut = _init_bn(outdim)
            blocks.append(b)
            indim = outdim
    # spatial size after the trunk decides the flattened feature count (the reference hard-codes indim,
    # backbone.py:280 -- wrong for '48_3_32'; we report the real size)
    s = (img_size + 2 * pert["conv1_padding"] - pert["conv1_kernel"]) // pert["conv1_stride"] + 1
    for b in blocks:
        s = (s + 2 - 3) // b.C1.stride + 1
    s_out = s // pert["avgpool_kernel"]
    feat = indim * s_out * s_out
    cw = rng.normal(0.0, 1.0 / math.sqrt(feat), size=(num_classes, feat))
    return ResNetQ(name=name, in_channels=in_channels, img_size=img_size, bit_width=bit_width, conv1=conv1, bn1=_init_bn(list_out_dims[0]),
                   relu1=pert.get("relu1", True), blocks=blocks, avgpool_kernel=pert["avgpool_kernel"], final_feat_dim=feat,
                   classifier_w=cw, classifier_b=np.zeros(num_classes))


def ResNet20QAT(bit_width=4, in_channels=3, img_size=224, seed=0, num_classes=10):
    """reference backbone.py:319-331: channels [48, 56, 64], skip_single_downsample=True"""
    return build_resnet_q([3, 3, 3], [48, 56, 64], in_channels, img_size, bit_width, True, num_classes, seed, "ResNet20qat")


def ResNet18QAT(bit_width=4, in_channels=3, img_size=224, seed=0, num_classes=10):
    """reference backbone.py:334-344"""
    return build_resnet_q([2, 2, 2, 2], [64, 128, 256, 512], in_channels, img_size, bit_width, False, num_classes, seed, "ResNet18qat")


def tiny_resnet_q(in_channels=4, img_size=6, width=(6, 8), seed=0, bit_width=4):
    """Two-block miniature with the same block structure (identity and 1x1 shortcuts, a stride-2
    stage, floor-mode pooling): small enough for the CPU oracle's encrypted twin."""
    rng = np.random.default_rng(seed)
    conv1 = ConvLayer(_init_conv(rng, width[0], in_channels, 1), 1, 0)
    blocks = []
    indim = width[0]
    for i, outdim in enumerate(width):
        half = i == len(width) - 1
        b = QBlock(C1=ConvLayer(_init_conv(rng, outdim, indim, 3), 2 if half else 1, 1), BN1=_init_bn(outdim),
                   C2=ConvLayer(_init_conv(rng, outdim, outdim, 3), 1, 1), BN2=_init_bn(outdim))
        if indim != outdim:
            b.shortcut = ConvLayer(_init_conv(rng, outdim, indim, 1), 2 if half else 1, 0)
            b.BNshortcut = _init_bn(outdim)
        blocks.append(b)
        indim = outdim
    s = img_size
    for b in blocks:
        s = (s + 2 - 3) // b.C1.stride + 1
    k = s if s < 3 else s - 1                 # floor-mode pooling drops a border when s > k
    feat = indim * (s // k) ** 2
    return ResNetQ(name="tiny", in_channels=in_channels, img_size=img_size, bit_width=bit_width, conv1=conv1, bn1=_init_bn(width[0]), relu1=True,
                   blocks=blocks, avgpool_kernel=k, final_feat_dim=feat,
                   classifier_w=rng.normal(0, 1.0 / math.sqrt(feat), size=(10, feat)), classifier_b=np.zeros(10))


def float_forward(model, x):
    """Float evaluation of the trunk as the reference's float twin computes it (models/backbone.py:47-58,182-184): no
    quantisers, BatchNorm on its running statistics.  Used to check imported weights/topology against the reference."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64))

    def bn(b, h):
        return F.batch_norm(h, t(b.mean), t(b.var), t(b.gamma), t(b.beta), False, 0.0, b.eps)

    def conv(c, h):
        return F.conv2d(h, t(c.weight), stride=c.stride, padding=c.pad)

    h = bn(model.bn1, conv(model.conv1, t(x)))
    if model.relu1:
        h = F.relu(h)
    for b in model.blocks:
        o = bn(b.BN2, conv(b.C2, F.relu(bn(b.BN1, conv(b.C1, h)))))
        sc = h if b.shortcut is None else bn(b.BNshortcut, conv(b.shortcut, h))
        h = F.relu(o + sc)
    return F.avg_pool2d(h, model.avgpool_kernel).flatten(1).numpy()


def trunk_prefix(model, n_blocks, avgpool_kernel):
    """The stem and the first `n_blocks` residual blocks of `model`, closed with the usual pooling + output quantiser: a
    smaller circuit with the same layers, tiers and per-site precisions as the head of the full one (parity tests of
    configurations whose full-size encrypted run takes many minutes).  The input may be any spatial crop."""
    import dataclasses
    blocks = list(model.blocks[:n_blocks])
    feat = blocks[-1].C2.weight.shape[0] if blocks else model.conv1.weight.shape[0]
    return dataclasses.replace(model, name=model.name + f"[:{n_blocks}]", blocks=blocks, avgpool_kernel=avgpool_kernel, final_feat_dim=feat,
                               classifier_w=None, classifier_b=None)


model_dict = dict(ResNet20qat=ResNet20QAT, ResNet18qat=ResNet18QAT)   # reference io_utils.py:5-10 (QAT entries)
