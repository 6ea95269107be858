"""The streaming kernels of the levelled operators (residual `torch.add` backbone.py:102, `nn.AvgPool2d(k)` :276 as a
window sum, and the shift / offset copy that opens a rounding chain) ONE AT A TIME through the C ABI, on rows stored at mixed
effective dimensions -- the storage form of a session's tensors -- against numpy and the oracle's ref_sum_pool.
Integer wrap-around arithmetic: bit-exact.  (Inside circuits these kernels are covered by every encrypted run; this is the stand-alone
check the round-2 review asked for.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rows(rng, count, dim, deff):
    """random rows of dim mask words + body; what lies between deff and the body is GARBAGE on purpose: the kernels must not read it"""
    r = rng.integers(0, 2 ** 64, (count, dim + 1), dtype=np.uint64)
    return r


def dense(r, deff, dim_o):
    """the ciphertext a stored row stands for, widened to dim_o mask words: words from deff on are zero"""
    out = np.zeros((r.shape[0], dim_o + 1), np.uint64)
    out[:, :deff] = r[:, :deff]
    out[:, dim_o] = r[:, -1]
    return out


@pytest.mark.parametrize("dim_a,deff_a,dim_b,deff_b,dim_o", [(64, 64, 64, 64, 64), (96, 40, 32, 32, 64), (16, 0, 200, 130, 130), (33, 33, 7, 5, 50)])
def test_add_rows_mixed_effective_dimensions(gpu_ctx, dim_a, deff_a, dim_b, deff_b, dim_o):
    rng = np.random.default_rng(dim_a * 7 + dim_b)
    a, b = rows(rng, 37, dim_a, deff_a), rows(rng, 37, dim_b, deff_b)
    got = gpu_ctx.add_rows(a, deff_a, b, deff_b, dim_o)
    assert np.array_equal(got, dense(a, deff_a, dim_o) + dense(b, deff_b, dim_o))


def test_add_rows_refuses_a_row_that_cannot_hold_the_sum(gpu_ctx):
    from dctfhe._lib import DctfheError
    rng = np.random.default_rng(1)
    with pytest.raises(DctfheError, match="cannot hold the sum"):
        gpu_ctx.add_rows(rows(rng, 2, 64, 64), 64, rows(rng, 2, 64, 8), 8, 32)
    with pytest.raises(DctfheError, match="effective dimension"):
        gpu_ctx.add_rows(rows(rng, 2, 16, 16), 17, rows(rng, 2, 16, 16), 16, 32)


@pytest.mark.parametrize("dim_a,deff_a,dim_o,nwords,shift", [(64, 64, 64, 64, 0), (48, 20, 100, 32, 5), (48, 48, 100, 48, 57), (10, 10, 40, 40, 3), (30, 12, 30, 0, 9)])
def test_affine_rows_touch_what_they_should_and_nothing_else(gpu_ctx, dim_a, deff_a, dim_o, nwords, shift):
    """first nwords mask words = a << shift (zero from deff on), body = (body << shift) + offset, the words in between untouched"""
    rng = np.random.default_rng(dim_o + shift)
    a = rows(rng, 23, dim_a, deff_a)
    before = rows(rng, 23, dim_o, dim_o)
    add = int(rng.integers(0, 2 ** 63)) * 2 + 1
    got = gpu_ctx.affine_rows(a, deff_a, before, nwords, shift, add)
    want = before.copy()
    src = dense(a, deff_a, max(dim_o, dim_a))
    want[:, :nwords] = src[:, :nwords] << np.uint64(shift)
    want[:, dim_o] = (a[:, -1] << np.uint64(shift)) + np.uint64(add)
    assert np.array_equal(got, want)
    assert np.array_equal(got[:, nwords:dim_o], before[:, nwords:dim_o])


@pytest.mark.parametrize("C,H,W,K,dim,deff,dim_o", [(3, 8, 8, 7, 40, 40, 40), (5, 4, 4, 3, 64, 24, 32), (2, 6, 9, 2, 17, 9, 9), (4, 7, 7, 7, 12, 12, 20)])
def test_sum_pool_rows_against_numpy_and_the_oracle(gpu_ctx, oracle, C, H, W, K, dim, deff, dim_o):
    """window sums with floor semantics: 8x8 with K = 7 keeps rows / columns 0..6 (ResNet-20, backbone.py:400), 4x4 with K = 3 the top-left
    3x3 (ResNet-18 3x32^2, :445)"""
    rng = np.random.default_rng(C * 100 + H)
    B = 2
    x = rng.integers(0, 2 ** 64, (B, C, H, W, dim + 1), dtype=np.uint64)
    got = gpu_ctx.sum_pool_rows(x, deff, K, dim_o)
    Ho, Wo = H // K, W // K
    xd = np.zeros((B, C, H, W, dim_o + 1), np.uint64)
    xd[..., :deff] = x[..., :deff]
    xd[..., dim_o] = x[..., -1]
    want = xd[:, :, :Ho * K, :Wo * K].reshape(B, C, Ho, K, Wo, K, dim_o + 1).sum(axis=(3, 5), dtype=np.uint64)
    assert got.shape == want.shape and np.array_equal(got, want)
    # the oracle's own pooling on the widened ciphertexts (one image at a time)
    import ctypes
    u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
    for b in range(B):
        ref = np.zeros((C, Ho, Wo, dim_o + 1), np.uint64)
        oracle.lib().ref_sum_pool(np.ascontiguousarray(xd[b]), C, H, W, dim_o, K, ref)
        assert np.array_equal(got[b], ref)
