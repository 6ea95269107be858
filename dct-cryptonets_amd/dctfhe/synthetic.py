"""Synthetic DCT-packed CIFAR-10 batches: seeded uint8 images pushed through the
restated plaintext front-end (dctfhe/frontend.py).  There is no network for the real dataset."""
import numpy as np

from . import frontend


def synthetic_images(batch, seed=42, size=32):
    return np.random.default_rng(seed).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)


def synthetic_dct_batch(batch, seed=42, filter_size=4, image_size_dct=16, channels=24):
    """float32 [B, channels, S, S] exactly as the reference Dataset transform would hand it to forward()"""
    imgs = synthetic_images(batch, seed)
    tf = frontend.dct_eval_transform(filter_size=filter_size, image_size_dct=image_size_dct, channels=channels)
    return np.stack([tf(im) for im in imgs]).astype(np.float32)


def centre_classifier(model, feats_calib):
    """Random-init trunks map every synthetic image to nearly the same feature vector, so a zero-bias random classifier
    (reference utils.py:22 `nn.Linear`, checkpoints are not shipped) predicts one label for all of them and a label
    comparison checks nothing.  Give the seeded classifier the bias that centres its logits on the calibration
    features: labels then follow each image's own deviation.  -> bias [classes]"""
    w = np.asarray(model.classifier_w, np.float64)
    model.classifier_b = -(w @ np.asarray(feats_calib, np.float64).mean(axis=0))
    return model.classifier_b
