"""P4: the restated front-end against goldens captured from the reference's own functions
(tools/make_goldens.py -> tests/golden/frontend_golden.npz)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz"))


def test_stats_table_matches_reference():
    from dctfhe import frontend
    mean, std = frontend.load_stats()
    assert mean.shape == std.shape == (192,)
    assert np.array_equal(mean, G["stats_mean"]) and np.array_equal(std, G["stats_std"])
    assert np.allclose(mean[:3], [-88.397, 0.014242, 0.0034117], rtol=1e-4)      # SURVEY appendix E spot values


@pytest.mark.parametrize("plane,size,key", [("plane64", 4, "dct4_plane64"), ("plane64", 8, "dct8_plane64"), ("plane_odd", 4, "dct4_plane_odd")])
def test_matrix2dct(plane, size, key):
    from dctfhe import frontend
    got = frontend.matrix2dct(G[plane], size)
    assert got.shape == G[key].shape
    assert np.allclose(got, G[key], rtol=0, atol=1e-9)


def test_matrix2dct_dc_term():
    from dctfhe import frontend
    y = G["plane64"]
    d = frontend.matrix2dct(y, 4)
    assert np.isclose(d[0, 0, 0], (y[:4, :4].astype(np.int32) - 128).sum() / 4.0)


@pytest.mark.parametrize("tag,ch,filt", [("c24f4", 24, 4), ("c48f8", 48, 8), ("c48f4", 48, 4)])
def test_subset_aggregate_normalize(tag, ch, filt):
    from dctfhe import frontend
    sy, scb, scr = frontend.subset_indices(ch, "default", filt)
    assert list(G[f"{tag}_subset_y"]) == list(sy) and list(G[f"{tag}_subset_cb"]) == list(scb) and list(G[f"{tag}_subset_cr"]) == list(scr)
    assert list(G[f"{tag}_norm_subset"]) == frontend.normalize_indices(ch)
    out = frontend.subset_aggregate_normalize(G[f"{tag}_y"], G[f"{tag}_cb"], G[f"{tag}_cr"], ch, "default", filt)
    assert out.dtype == np.float32 and out.shape == G[f"{tag}_out"].shape
    assert np.allclose(out, G[f"{tag}_out"], rtol=1e-6, atol=1e-6)


def test_normalize_index_quirk():
    """filter 4 selects with the 4x4 table but normalises with the 8x8 default indices"""
    from dctfhe import frontend
    assert frontend.normalize_indices(24) == [0, 1, 2, 3, 4, 5, 8, 9, 10, 16, 17, 18, 24, 32, 64, 65, 67, 72, 88, 128, 129, 131, 136, 152]
    assert frontend.subset_indices(24, "default", 4)[0] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13]


def test_full_transform_self_consistency():
    """cv2-dependent stages are unpinned (parity unpinned: OpenCV absent); check shape, dtype, determinism, crop geometry"""
    from dctfhe import frontend, synthetic
    x = synthetic.synthetic_dct_batch(3, seed=42)
    assert x.shape == (3, 24, 16, 16) and x.dtype == np.float32 and np.isfinite(x).all()
    assert np.array_equal(x, synthetic.synthetic_dct_batch(3, seed=42))
    img = synthetic.synthetic_images(1, 42)[0]
    up = frontend.resize_u8(img, 73, 73)
    assert up.shape == (73, 73, 3) and frontend.center_crop(up, 64).shape == (64, 64, 3)
    assert np.array_equal(frontend.center_crop(up, 64), up[4:68, 4:68])          # round(4.5) == 4 (banker's)
    flat = np.full((8, 8, 3), 128, np.uint8)
    y, cr, cb = frontend.rgb_to_ycrcb_u8(flat)
    assert (y == 128).all() and (cr == 128).all() and (cb == 128).all()
    assert np.array_equal(frontend.halve_u8(np.array([[1, 2], [3, 4]], np.uint8)), [[3]])


def test_jpeg_domain_dct_self_consistency():
    """8x8 path (reference cvfunctional.py:21-26) -- PARITY UNPINNED (TurboJPEG / jpeg2dct absent): checks the JPEG
    definitions the restatement follows, not the libraries."""
    from dctfhe import frontend
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    y, cb, cr = frontend.transform_dct_jpeg(img)
    assert y.shape == (4, 6, 64) and cb.shape == (2, 3, 64) and cr.shape == (2, 3, 64)
    assert np.array_equal(y, np.round(y)) and np.abs(y).max() <= 1024
    # a flat grey image: only DC, luma DC = 8 * (Y - 128), chroma DC = 0 (Cb = Cr = 128)
    flat = np.full((16, 16, 3), 200, np.uint8)
    fy, fcb, fcr = frontend.transform_dct_jpeg(flat)
    assert fy[0, 0, 0] == 8 * (200 - 128) and not fy[..., 1:].any() and not fcb.any() and not fcr.any()
    # the encoder reads the RGB array as BGR: a pure "red" array is blue inside the JPEG -> Cb high, Cr low
    red = np.zeros((16, 16, 3), np.uint8); red[..., 0] = 255
    _, rcb, rcr = frontend.transform_dct_jpeg(red)
    assert rcb[0, 0, 0] > 0 > rcr[0, 0, 0]
    # inverse DCT of the quantised coefficients reproduces the luma plane within the rounding of 64 coefficients
    T = np.array([[1 / np.sqrt(8) if i == 0 else np.sqrt(2 / 8) * np.cos((2 * j + 1) * i * np.pi / 16) for j in range(8)] for i in range(8)])
    r, g, b = [img[..., i].astype(np.int64) for i in (2, 1, 0)]
    luma = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    rec = np.einsum("ij,abjk,kl->abil", T.T, y.reshape(4, 6, 8, 8), T) + 128
    assert np.abs(rec.transpose(0, 2, 1, 3).reshape(32, 48) - luma).max() < 4.0
    with pytest.raises(ValueError):
        frontend.transform_dct_jpeg(np.zeros((20, 16, 3), np.uint8))


def test_filter8_eval_transform_shapes():
    from dctfhe import frontend
    img = np.random.default_rng(2).integers(0, 256, (300, 260, 3), dtype=np.uint8)
    for ch in (48, 64):
        x = frontend.dct_eval_transform(filter_size=8, image_size_dct=14, channels=ch)(img)
        assert x.shape == (ch, 14, 14) and x.dtype == np.float32 and np.isfinite(x).all()


def test_subset_patterns_square_triangle_learned():
    """every SubsetDCT pattern of the reference (cvtransforms.py:117-128) from the captured tables; the two spelled-out tables agree with
    the captured ones; NormalizeDCT keeps indexing the statistics with the DEFAULT table whatever the pattern (datamgr.py:209-216)"""
    from dctfhe import frontend
    T = frontend.subset_tables()
    assert set(T) == {"filter4", "default", "square", "learned", "triangle"}
    for ch, v in frontend.SUBSET_DEFAULT.items():
        assert tuple(map(list, v)) == T["default"][ch]
    for ch, v in frontend.SUBSET_FILTER4.items():
        assert tuple(map(list, v)) == T["filter4"][ch]
    for pattern, ch, sizes in [("square", 24, (16, 4, 4)), ("triangle", 24, (12, 6, 6)), ("learned", 24, (14, 5, 5)), ("triangle", 48, (28, 10, 10)),
                               ("square", 64, (44, 10, 10)), ("default", 32, (22, 5, 5))]:
        y, cb, cr = frontend.subset_indices(ch, pattern, 8)
        assert (len(y), len(cb), len(cr)) == sizes and len(y) + len(cb) + len(cr) == ch
        assert all(0 <= i < 64 for i in y + cb + cr) and len(set(y)) == len(y)
    assert frontend.subset_indices(24, "square", 4) == frontend.subset_indices(24, "default", 4)       # filter 4 ignores the pattern
    # whole transform with a non-default pattern: selection by the pattern's table, statistics by the default one
    rng = np.random.default_rng(5)
    planes = [rng.normal(0, 30, (64, 6, 6)).astype(np.float32) for _ in range(3)]
    out = frontend.subset_aggregate_normalize(*planes, channels=24, pattern="triangle", filter_size=8)
    y, cb, cr = frontend.subset_indices(24, "triangle", 8)
    mean, std = frontend.load_stats()
    idx = frontend.normalize_indices(24)
    want = (np.concatenate([planes[0][y], planes[1][cb], planes[2][cr]]) - mean[idx].astype(np.float32)[:, None, None]) / std[idx].astype(np.float32)[:, None, None]
    assert out.shape == (24, 6, 6) and np.array_equal(out, want)
    with pytest.raises(ValueError, match="no 'learned' coefficient table"):
        frontend.subset_indices(48, "learned", 8)
    with pytest.raises(ValueError, match="dct_pattern"):
        frontend.subset_indices(24, "zigzag", 8)
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=4, channels=24, dct_pattern="square")
    assert tf(rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)).shape == (24, 4, 4)


def test_jpeg_integer_dct_restatement():
    """libjpeg's integer forward DCT + quality-100 quantiser (frontend.jpeg_quantised_dct; PARITY UNPINNED -- TurboJPEG / jpeg2dct absent):
    integers only; within libjpeg's documented accuracy of the exact DCT (the scaled-by-8 result is good to about one unit, i.e. 1/8 here,
    before the quantiser rounds); exact on constant blocks; linear in the DC term."""
    from dctfhe import frontend
    rng = np.random.default_rng(9)
    pl = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    q = frontend.jpeg_quantised_dct(pl)
    assert q.dtype == np.int64 and q.shape == (6, 8, 64)
    exact = frontend.matrix2dct(pl, 8)
    assert np.abs(q - exact).max() <= 0.5 + 0.15                      # rounding + the fixed-point transform's own error
    assert (q == frontend._round_half_away(exact)).mean() > 0.85      # ... so most coefficients are the rounded exact ones, not all
    flat = np.full((8, 8), 200, np.uint8)
    qf = frontend.jpeg_quantised_dct(flat)[0, 0]
    assert qf[0] == (200 - 128) * 8 and not qf[1:].any()              # DC = 8 * mean for the orthonormal 8x8 DCT, AC = 0
    # DC-term ties: a block mean with fraction 1/16 puts the exact DC on x.5; the integer path rounds half away from zero, both signs
    for v, want in [(129, 8), (127, -8)]:
        blk = np.full((8, 8), 128, np.uint8)
        blk[0, :4] = v                                                # sum of (p - 128) = +-4 -> DC = +-0.5 -> +-1
        assert frontend.jpeg_quantised_dct(blk)[0, 0, 0] == (1 if v > 128 else -1)
    a, b, c = frontend.transform_dct_jpeg(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8))
    assert a.shape == (4, 4, 64) and b.shape == c.shape == (2, 2, 64) and np.array_equal(a, np.rint(a))
