"""Homomorphic evaluation driver with the reference's command line (drop-in for run_homomorphic_eval.sh).

Mirrors reference dct-cryptonets/homomorphic_eval.py:89-443 and the flag set of io_utils.py:13-90 (same names,
defaults and choices) on top of the dctfhe engine.  Differences forced by the environment: datasets cannot be
downloaded (homomorphic_eval.py:139-142 needs the network), so images are synthetic and seeded
(dctfhe/synthetic.py) with random labels; --checkpoint_path
is read by dctfhe/checkpoint.py (weights, BatchNorm statistics, classifier; activation scales are re-calibrated) and a
missing file takes the reference's own "random weights" branch (homomorphic_eval.py:254-256) with the same warning.
Printed lines keep the reference's wording so logs stay comparable.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dctfhe import frontend, models, synthetic                                   # noqa: E402
from dctfhe.quantized_module import Configuration, compile_brevitas_qat_model, compile_torch_model   # noqa: E402


def parse_args():
    """reference io_utils.py:13-90, script == 'homomorphic_eval'"""
    parser = argparse.ArgumentParser(description="DCT-CryptoNets (Homomorphic Evaluation)", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    g = parser.add_argument_group("Default arguments")
    g.add_argument("--dataset", default="cifar10", choices=["cifar10", "ImageNet"], help="Choose image dataset")
    g.add_argument("--model", default="ResNet18qat", choices=["ResNet20", "ResNet20qat", "ResNet18", "ResNet18qat"], help="Choose model architecture")
    g.add_argument("--num_classes", default=10, type=int, help="Number of prediction classes")
    g.add_argument("--dataset_path", metavar="PATH", help="Path to directory with dataset")
    g.add_argument("--save_path", metavar="PATH", help="Path to parent directory to save checkpoints")
    g.add_argument("--train_aug", action="store_true", help="Perform data augmentation during training? (flag)")
    g.add_argument("--dct_status", action="store_true", help="Is this a DCT-based model? (flag)")
    g.add_argument("--channels", default=64, type=int, choices=[3, 6, 24, 48, 64, 192], help="top-n low-frequency DCT components (3 for RGB)")
    g.add_argument("--filter_size", default=8, type=int, help="DCT filter size")
    g.add_argument("--image_size", default=32, type=int, help="Size of non-DCT spatial dimensions")
    g.add_argument("--image_size_dct", default=56, type=int, help="Size of DCT spatial dimensions")
    g.add_argument("--dct_pattern", default="default", type=str, choices=["default", "square", "triangle", "learned"], help="DCT subset pattern")
    g.add_argument("--bit_width", default=4, type=int, help="Quantization bit-width")
    g.add_argument("--dropout", default=None, type=float, help="Fraction of fc layer to dropout")
    g.add_argument("--verbose", default=True, type=bool, help="Verbose log outputs")
    h = parser.add_argument_group("Homomorphic evaluation arguments")
    h.add_argument("--checkpoint_path", type=str, help="Filepath to checkpoint")
    h.add_argument("--calib_batch_size", default=64, type=int, help="Batch size used for post-training quantization calibration")
    h.add_argument("--test_batch_size", default=1, type=int, help="Inference batch size")
    h.add_argument("--test_subset", default=1, type=int, help="Number of images to perform inference on")
    h.add_argument("--fhe_mode", default="simulate", type=str, choices=["simulate", "execute"], help="simulate (accuracy) or execute (latency)")
    h.add_argument("--rounding_threshold_bits", default=6, type=int, help="Scaling factor to remove least significant bits")
    h.add_argument("--n_bits", default=5, type=int, help="Bit-width of homomorphic circuit")
    h.add_argument("--p_error", default=0.01, type=float, help="PBS error probability")
    h.add_argument("--reliability_test", default=True, help="Perform accuracy reliability analysis over random subsets?")
    e = parser.add_argument_group("dctfhe additions")
    e.add_argument("--seed", default=42, type=int, help="seed of the synthetic images")
    e.add_argument("--device", default=0, type=int, help="GPU index")
    e.add_argument("--rounding_method", default="exact", choices=["exact", "approximate"],
                   help="exact = the reference's call (int rounding_threshold_bits); approximate = its README's suggested speed-up")
    e.add_argument("--tier_policy", default="exact", choices=["exact", "p_error"],
                   help="exact: outputs equal the integer circuit whatever --p_error; p_error: cheaper tiers failing with probability <= --p_error per look-up")
    return parser.parse_args()


class AverageMeter:            # reference utils.py:74-89
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):     # reference utils.py:111-124
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def test_encrypted(params, q_module, batches, fhe_mode, cls_w, cls_b):
    """reference homomorphic_eval.py:60-86"""
    top1, top5 = AverageMeter(), AverageMeter()
    for data, target in batches:
        encoder_output = q_module.forward(data, fhe=fhe_mode)                                      # :70
        output = torch.from_numpy(encoder_output).float() @ cls_w.T + cls_b                        # clear classifier, :73-76
        prec1, prec5 = accuracy(output, target, topk=(1, min(5, output.shape[1])))
        top1.update(prec1.item(), data.shape[0])
        top5.update(prec5.item(), data.shape[0])
    return top1, top5


def main():
    params = parse_args()
    quantization_type = "QAT" if "qat" in str(params.model).lower() else "PTQ"             # :95-98
    if params.dct_status:
        tf = frontend.dct_eval_transform(params.filter_size, params.image_size_dct, params.channels, params.dct_pattern)
        in_ch, img = params.channels, params.image_size_dct
    else:
        tf = frontend.rgb_eval_transform(params.image_size)
        in_ch, img = 3, params.image_size

    def make(n, seed):
        imgs = synthetic.synthetic_images(n, seed)
        x = np.stack([tf(im) for im in imgs]).astype(np.float32)
        y = torch.from_numpy(np.random.default_rng(seed + 1).integers(0, params.num_classes, n))
        return x, y

    name = params.model if params.model.endswith("qat") else params.model + "qat"
    model = models.model_dict[name](bit_width=params.bit_width, in_channels=in_ch, img_size=img, num_classes=params.num_classes)
    if params.checkpoint_path and os.path.isfile(params.checkpoint_path):                     # :247-253
        from dctfhe import checkpoint
        meta, unused = checkpoint.load_checkpoint(params.checkpoint_path, model)
        print(f"Loaded checkpoint {params.checkpoint_path} (epoch {meta.get('epoch')}, prec1 {meta.get('prec1')}); "
              f"{len(unused)} quantiser-scale entries re-derived from the calibration batch")
    else:
        print("WARNING: No checkpoint loaded. Using random weights (for testing only)")       # :254-256
        print("Results will NOT be meaningful!")

    calib_data, _ = make(params.calib_batch_size, params.seed + 100)
    print("\nCompiling FHE Model (this can take up to 10 minutes for larger networks)...")
    configuration = Configuration(show_progress=False, progress_tag=True, progress_title="Evaluation: ")
    t = time.time()
    compile_fn = compile_brevitas_qat_model if quantization_type == "QAT" else compile_torch_model
    rtb = params.rounding_threshold_bits if params.rounding_method == "exact" else {"n_bits": params.rounding_threshold_bits, "method": "approximate"}
    q_module = compile_fn(model, calib_data, rounding_threshold_bits=rtb, n_bits=params.n_bits, p_error=params.p_error,
                          configuration=configuration, verbose=params.verbose, device=params.device, tier_policy=params.tier_policy)
    print(f"Time for FHE compilation {time.time() - t:.2f}")
    bitwidth = q_module.fhe_circuit.graph.maximum_integer_bit_width()
    print(f"Max bit-width: {bitwidth} bits" + (" -> it works in FHE!!" if bitwidth <= 16 else " too high for FHE computation"))
    if params.verbose:
        with open("mlir.txt", "a") as f:
            print(q_module.fhe_circuit.mlir, file=f)
    t = time.time()
    q_module.fhe_circuit.keygen()
    print(f"Keygen time: {time.time() - t:.2f}s")

    x, y = make(params.test_subset, params.seed)
    bs = params.test_batch_size
    batches = [(x[i:i + bs], y[i:i + bs]) for i in range(0, len(x), bs)]
    cls_w, cls_b = torch.from_numpy(model.classifier_w).float(), torch.from_numpy(model.classifier_b).float()

    print(f"\nRunning UNENCRYPTED model on a subset of {params.test_subset} images...")
    top1, top5 = test_encrypted(params, q_module, batches, "disable", cls_w, cls_b)
    print(f"[Test] Top-1 Acc: {top1.avg:.3f}% | Top-5 Acc: {top5.avg:.3f}%")

    t = time.time()
    print(f"\nRunning ENCRYPTED test inference in {params.fhe_mode.upper()} mode on a subset of {params.test_subset} images...")
    top1, top5 = test_encrypted(params, q_module, batches, params.fhe_mode, cls_w, cls_b)
    time_per_inference = (time.time() - t) / params.test_subset
    print(f"[Test] Top-1 Acc: {top1.avg:.3f}% | Top-5 Acc: {top5.avg:.3f}% | Time per inference in FHE: {time_per_inference:.2f}")

    # reliability analysis over random subsets (reference homomorphic_eval.py:366-440: random states 27 and 28, simulate only).
    # `simulate` samples the compiler's noise model at every look-up; with the default exact-evaluation tiers (expected
    # failing look-ups per image ~1e-7) that coincides with the clear circuit, with --tier_policy p_error / --rounding_method
    # approximate it reproduces the reference-style stochastic regime.
    if params.reliability_test is not None and params.fhe_mode == "simulate":
        print("\n============ Encrypted Reliability Analysis ============")
        top1_plain, top5_plain, top1_enc, top5_enc = [], [], [], []
        for rstate in range(27, 29):
            print(f"\n\nRunning ENCRYPTED test inference on subset of {params.test_subset} with random state {rstate}...")
            xr, yr = make(params.test_subset, rstate)
            br = [(xr[i:i + bs], yr[i:i + bs]) for i in range(0, len(xr), bs)]
            p1, p5 = test_encrypted(params, q_module, br, "disable", cls_w, cls_b)        # clear integer circuit
            top1_plain.append(p1.avg); top5_plain.append(p5.avg)
            print(f"[Test] UNENCRYPTED Top-1 Acc: {p1.avg:.3f}% | Top-5 Acc: {p5.avg:.3f}%")
            t = time.time()
            e1, e5 = test_encrypted(params, q_module, br, params.fhe_mode, cls_w, cls_b)
            print(f"[Test] ENCRYPTED Top-1 Acc: {e1.avg:.3f}% | Top-5 Acc: {e5.avg:.3f}% | "
                  f"Time per inference in FHE: {(time.time() - t) / params.test_subset:.2f}")
            top1_enc.append(e1.avg); top5_enc.append(e5.avg)
        print("\n--------Encrypted Reliability Analysis Results--------")
        print(f"Unencrypted top1 acc: {top1_plain}")
        print(f"Unencrypted top5 acc: {top5_plain}")
        print(f"Encrypted top1 acc: {top1_enc}")
        print(f"Encrypted top5 acc: {top5_enc}")
        print("--------------------------------------------------------")
    print("Done")
    q_module.close()


if __name__ == "__main__":
    try:
        main()
    except KeyboardInterrupt:
        print("Interrupted")
        os._exit(130)
