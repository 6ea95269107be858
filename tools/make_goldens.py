"""Generates tests/golden/*.npz from the reference's own pure-numpy / pure-torch functions.

Runs only in the build container (it reads /root/reference; the GPU box never does).  Recipe as in
SURVEY.md Appendix E: inert stand-ins are registered for the *absent* third-party packages
(cv2, turbojpeg, jpeg2dct, brevitas) so that module-level imports succeed; any reference function
that would really call one of them is NOT used as an oracle.  What is captured:
  - data.train_upscaled_static_mean/std          (192-entry data constants, data/__init__.py:289,329)
  - cvtransforms.subset_channel_index*           (index tables of the five SubsetDCT patterns, data constants)
  - cvfunctional.matrix2dct on seeded planes     (cvfunctional.py:37-57)
  - cvtransforms.SubsetDCT/Aggregate/NormalizeDCT on seeded tensors (cvtransforms.py:117-208)
  - the float twin models.backbone.ResNet20/ResNet18: conv output shapes, parameter counts
  - the same twin under seeded weights (dctfhe.torch_import.seed_parameters): state-dict key names/shapes and the float
    forward output on a seeded input
Outputs are data only (inputs + expected outputs).
"""
import importlib
import os
import sys
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/dct-cryptonets"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    for name in ("cv2", "jpeg2dct", "jpeg2dct.numpy", "turbojpeg", "brevitas", "brevitas.nn", "brevitas.quant", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = MagicMock()
    sys.path.insert(0, REF)
    import torch
    data = importlib.import_module("data")
    Fn = importlib.import_module("data.cvfunctional")
    Tr = importlib.import_module("data.cvtransforms")

    mean = np.array(data.train_upscaled_static_mean, np.float64)
    std = np.array(data.train_upscaled_static_std, np.float64)
    np.savez(os.path.join(ROOT, "dct-cryptonets_amd", "dctfhe", "data", "dct_stats.npz"), mean=mean, std=std)
    # the kept-coefficient index tables of SubsetDCT / NormalizeDCT for every pattern (data constants, cvtransforms.py:1600-1860)
    import json
    tables = {"filter4": Tr.subset_channel_index_filtersize_4, "default": Tr.subset_channel_index, "square": Tr.subset_channel_index_square,
              "learned": Tr.subset_channel_index_learned, "triangle": Tr.subset_channel_index_triangle}
    with open(os.path.join(ROOT, "dct-cryptonets_amd", "dctfhe", "data", "subset_tables.json"), "w") as f:
        json.dump({name: {str(ch): [list(map(int, part)) for part in v] for ch, v in d.items()} for name, d in tables.items()}, f, separators=(",", ":"))

    g = {"stats_mean": mean, "stats_std": std}
    rng = np.random.default_rng(0)
    y64 = rng.integers(0, 256, size=(64, 64), dtype=np.uint8)
    g["plane64"] = y64
    g["dct4_plane64"] = Fn.matrix2dct(y64, 4)
    g["dct8_plane64"] = Fn.matrix2dct(y64, 8)
    odd = rng.integers(0, 256, size=(30, 21), dtype=np.uint8)     # trailing rows/cols dropped
    g["plane_odd"] = odd
    g["dct4_plane_odd"] = Fn.matrix2dct(odd, 4)

    for (ch, filt, tag) in [(24, 4, "c24f4"), (48, 8, "c48f8"), (48, 4, "c48f4"), (64, 8, "c64f8")]:
        nf = 64 if filt == 8 else 16
        S = 6
        ty = torch.from_numpy(rng.normal(0, 50, (nf, S, S))).float()
        tcb = torch.from_numpy(rng.normal(0, 20, (nf, S, S))).float()
        tcr = torch.from_numpy(rng.normal(0, 20, (nf, S, S))).float()
        sub = Tr.SubsetDCT(channels=ch, pattern="default", filter_size=filt)
        agg = Tr.Aggregate()(sub((ty, tcb, tcr)))
        norm = Tr.NormalizeDCT(data.train_upscaled_static_mean, data.train_upscaled_static_std, channels=ch)
        out = norm(agg.clone())[0]
        g[f"{tag}_y"], g[f"{tag}_cb"], g[f"{tag}_cr"] = ty.numpy(), tcb.numpy(), tcr.numpy()
        g[f"{tag}_subset_y"] = np.array(sub.subset_y)
        g[f"{tag}_subset_cb"] = np.array(sub.subset_cb)
        g[f"{tag}_subset_cr"] = np.array(sub.subset_cr)
        g[f"{tag}_norm_subset"] = np.array(norm.subset)
        g[f"{tag}_agg"] = agg.numpy()
        g[f"{tag}_out"] = out.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frontend_golden.npz"), **g)

    # float twin topology
    bb = importlib.import_module("models.backbone")
    topo = {}
    for (fn, cin, size, tag) in [(bb.ResNet20, 24, 16, "r20_24_16"), (bb.ResNet20, 3, 32, "r20_3_32"), (bb.ResNet18, 3, 32, "r18_3_32"),
                                 (bb.ResNet18, 48, 112, "r18_48_112")]:
        m = fn(in_channels=cin, img_size=size)
        m.eval()
        shapes = []
        hooks = []
        for name, mod in m.named_modules():
            if isinstance(mod, torch.nn.Conv2d):
                hooks.append(mod.register_forward_hook(
                    lambda md, inp, out, name=name: shapes.append((md.in_channels, md.out_channels, md.kernel_size[0], md.stride[0], md.padding[0],
                                                                   inp[0].shape[2], inp[0].shape[3], out.shape[2], out.shape[3]))))
        with torch.no_grad():
            o = m(torch.zeros(1, cin, size, size))
        for h in hooks:
            h.remove()
        topo[f"{tag}_convs"] = np.array(shapes, np.int64)
        topo[f"{tag}_out"] = np.array(o.shape, np.int64)
        topo[f"{tag}_params"] = np.array(sum(p.numel() for p in m.parameters()), np.int64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "topology_golden.npz"), **topo)

    # the float twin under seeded weights: state-dict naming + forward output (tests/test_torch_import.py)
    sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
    from dctfhe.torch_import import seed_parameters
    ti = {}
    for (fn, cin, size, tag) in [(bb.ResNet20, 24, 16, "r20_24_16"), (bb.ResNet18, 3, 32, "r18_3_32")]:
        m = seed_parameters(fn(in_channels=cin, img_size=size), 11)
        m.eval()
        x = np.random.default_rng(12).normal(0, 1, (2, cin, size, size)).astype(np.float32)
        with torch.no_grad():
            y = m(torch.from_numpy(x))
        sd = m.state_dict()
        ti[f"{tag}_keys"] = np.array(list(sd.keys()))
        ti[f"{tag}_shapes"] = np.array([";".join(map(str, v.shape)) for v in sd.values()])
        ti[f"{tag}_x"], ti[f"{tag}_y"] = x, y.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "torch_import_golden.npz"), **ti)
    print("goldens written")


if __name__ == "__main__":
    main()
