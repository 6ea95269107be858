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
    """round_coeffs: integer coefficient planes (the filter-8 path's quantised JPEG coefficients), up-sampled and rounded to even"""
    from dctfhe import frontend
    from dctfhe._lib import DctfheError
    rng = np.random.default_rng(3)
    y = rng.integers(0, 256, (2, 32, 32), dtype=np.uint8)
    c = rng.integers(0, 256, (2, 16, 16), dtype=np.uint8)
    sy, scb, scr = frontend.subset_indices(48, "default", 8)
    out = gpu_ctx.dct_frontend(y, c, c, 8, (sy, scb, scr), np.zeros(48), np.ones(48), round_coeffs=True)
    assert out.shape == (2, 48, 4, 4) and np.array_equal(out, np.rint(out))

    def same_up_to_ties(got, unrounded, rounded):
        """equal wherever the unrounded value is not (numerically) on a rounding tie -- DC terms are multiples of 1/8, so
        exact .5 ties occur and two summation orders may fall on either side (libjpeg's integer DCT differs there too)"""
        tie = np.abs(np.abs(unrounded - np.floor(unrounded)) - 0.5) < 1e-6
        assert np.array_equal(got[~tie], rounded[~tie].astype(np.float32)) and np.abs(got - rounded).max() <= 1 and tie.mean() < 0.05

    raw_y = frontend.matrix2dct(y[0], 8).transpose(2, 0, 1)[sy]
    same_up_to_ties(out[0, :len(sy)], raw_y, frontend._round_half_away(raw_y))
    # chroma: compare on blocks where no coefficient of the 2x2 source grid sits on a tie
    cq = frontend._round_half_away(frontend.matrix2dct(c[0], 8))
    raw_c = frontend._bilinear(cq, 4, 4).transpose(2, 0, 1)[scb]
    got_c = out[0, len(sy):len(sy) + len(scb)]
    assert np.abs(got_c - np.rint(raw_c)).max() <= 1 and (got_c == np.rint(raw_c).astype(np.float32)).mean() > 0.95
    with pytest.raises(DctfheError, match="index out of range"):
        gpu_ctx.dct_frontend(y, c, c, 8, ([64], [], []), np.zeros(1), np.ones(1))
    with pytest.raises(DctfheError, match="bad geometry"):
        gpu_ctx.dct_frontend(y[:, :24, :24], c, c, 8, (sy, scb, scr), np.zeros(48), np.ones(48))
