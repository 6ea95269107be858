"""SURVEY K10 on the device (rows a11/a12): dctfhe_dct_frontend against the goldens captured from the reference's own
`matrix2dct` / `SubsetDCT` / `Aggregate` / `NormalizeDCT` (tests/golden/frontend_golden.npz) and against the numpy
front-end those goldens pin.  Tolerances: the reference computes the DCT in f64 and stores / normalises in f32; the kernel
does the same with its own summation order -> 1e-4 absolute on coefficients of magnitude <= 1e3 (f32 output), 1e-5 after
division by the statistics."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz"))


@pytest.mark.parametrize("fs,key", [(4, "dct4_plane64"), (8, "dct8_plane64")])
def test_blockwise_dct_matches_reference_golden(gpu_ctx, fs, key):
    y = G["plane64"][None]
    S = 64 // fs
    allc = np.arange(fs * fs, dtype=np.int32)
    none = np.zeros(0, np.int32)
    out = gpu_ctx.dct_frontend(y, y, y, fs, (allc, none, none), np.zeros(fs * fs), np.ones(fs * fs))
    want = G[key].transpose(2, 0, 1)[None]                      # [1, fs*fs, S, S]
    assert out.shape == want.shape == (1, fs * fs, S, S)
    assert np.abs(out - want).max() < 1e-4


def test_trailing_rows_and_columns_are_dropped(gpu_ctx):
    """reference matrix2dct: `//` block counts (cvfunctional.py:48-49); the ABI takes whole-block planes, so crop as it does"""
    odd = G["plane_odd"]                                        # 30 x 21
    sq = odd[:20, :20][None]                                    # 5 x 5 blocks of 4
    allc = np.arange(16, dtype=np.int32)
    none = np.zeros(0, np.int32)
    out = gpu_ctx.dct_frontend(sq, sq, sq, 4, (allc, none, none), np.zeros(16), np.ones(16))
    assert np.abs(out[0] - G["dct4_plane_odd"][:5, :5].transpose(2, 0, 1)).max() < 1e-4


@pytest.mark.parametrize("channels", [24, 48])
def test_device_front_end_equals_numpy_front_end(gpu_ctx, channels):
    """whole evaluation transform, filter 4: colour / resize / crop / halving on the host, the rest on the GPU"""
    from dctfhe import frontend, synthetic
    imgs = synthetic.synthetic_images(5, 42)
    got = frontend.device_dct_batch(gpu_ctx, imgs, filter_size=4, image_size_dct=16, channels=channels)
    tf = frontend.dct_eval_transform(filter_size=4, image_size_dct=16, channels=channels)
    want = np.stack([tf(im) for im in imgs])
    assert got.shape == want.shape == (5, channels, 16, 16) and got.dtype == np.float32
    assert np.abs(got - want).max() < 1e-5 * max(1.0, np.abs(want).max())


def test_jpeg_domain_planes_and_errors(gpu_ctx):
    """round_coeffs (block size 8): libjpeg's integer DCT + quantiser on the device -- integer coefficient planes EQUAL to the host
    restatement (frontend.jpeg_quantised_dct), ties included (round 2 ran a float DCT here and the test had to tolerate +-1 on ties);
    chroma grids up-sampled and rounded to even like the reference's int16 planes"""
    from dctfhe import frontend
    from dctfhe._lib import DctfheError
    rng = np.random.default_rng(3)
    y = rng.integers(0, 256, (2, 32, 32), dtype=np.uint8)
    c = rng.integers(0, 256, (2, 16, 16), dtype=np.uint8)
    y[1, :8, :8] = 128
    y[1, 0, :4] = 129                                                    # a DC term exactly on a rounding tie (+0.5)
    y[1, 8:16, :8] = 128
    y[1, 8, :4] = 127                                                    # ... and on -0.5
    sy, scb, scr = frontend.subset_indices(48, "default", 8)
    out = gpu_ctx.dct_frontend(y, c, c, 8, (sy, scb, scr), np.zeros(48), np.ones(48), round_coeffs=True)
    assert out.shape == (2, 48, 4, 4) and np.array_equal(out, np.rint(out))
    for b in range(2):
        want_y = frontend.jpeg_quantised_dct(y[b]).transpose(2, 0, 1)[sy]
        assert np.array_equal(out[b, :len(sy)], want_y.astype(np.float32))
        cq = frontend.jpeg_quantised_dct(c[b]).astype(np.float64)
        want_c = np.rint(frontend._bilinear(cq, 4, 4)).transpose(2, 0, 1)[scb]
        assert np.array_equal(out[b, len(sy):len(sy) + len(scb)], want_c.astype(np.float32))
    assert out[1, 0, 0, 0] == 1 and out[1, 0, 1, 0] == -1               # the two ties: half away from zero, as libjpeg's quantiser
    with pytest.raises(DctfheError, match="index out of range"):
        gpu_ctx.dct_frontend(y, c, c, 8, ([64], [], []), np.zeros(1), np.ones(1))
    with pytest.raises(DctfheError, match="bad geometry"):
        gpu_ctx.dct_frontend(y[:, :24, :24], c, c, 8, (sy, scb, scr), np.zeros(48), np.ones(48))


def test_device_front_end_filter8_equals_numpy_front_end(gpu_ctx):
    """config #5's transform (8x8 JPEG-domain DCT, 48 channels) with DCT / quantiser / subset / up-sampling / normalisation on the GPU:
    the same float32 tensor as the numpy path (every stage before the final (x - mean) / std is integer)"""
    from dctfhe import frontend, synthetic
    imgs = synthetic.synthetic_images(3, 43, size=48)
    got = frontend.device_dct_batch(gpu_ctx, imgs, filter_size=8, image_size_dct=8, channels=48)
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=8, channels=48)
    want = np.stack([tf(im) for im in imgs])
    assert got.shape == want.shape == (3, 48, 8, 8) and got.dtype == np.float32
    # (the integer planes are compared bit for bit in the test above; here one f32 (x - mean) / std on top: an ulp at most)
    assert np.abs(got - want).max() <= 2e-7 * max(1.0, np.abs(want).max())
