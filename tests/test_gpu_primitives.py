"""Parity of the HIP primitives (through the C ABI) against the CPU oracle, on small parameter sets
that the oracle finishes in seconds.  Integer kernels (encrypt mask/decrypt, key switch, conv) are
bit-exact; the bootstrap is compared on decrypted values and on the noise it leaves (an f64 FFT on
two machines cannot agree bit for bit: one flipped rounding in a gadget decomposition yields a
different, equivalent ciphertext).  PARITY UNPINNED by the reference (oracle/tfhe_ref.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIERS_SMALL = [
    dict(n=40, k=1, logN=9, l=2, beta=12, lk=4, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -45),   # table tier
    dict(n=32, k=2, logN=8, l=2, beta=10, lk=3, betak=4, lwe_sigma=2.0 ** -22, glwe_sigma=2.0 ** -45),   # bit tier
    dict(n=40, k=1, logN=10, l=1, beta=20, lk=4, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -50, ksk_share=0),
]
D_SMALL = 1024


@pytest.fixture(scope="module")
def keys(gpu_ctx):
    from dctfhe.engine import Keys, make_params
    k = Keys(gpu_ctx, make_params(D_SMALL, 40, TIERS_SMALL, 2.0 ** -50), seed=7)
    yield k
    k.close()


def _centered(x):
    return x.astype(np.int64).astype(np.float64) / 2.0 ** 64


def test_encrypt_decrypt_matches_oracle(keys, oracle):
    S, s = keys.export_secret()
    assert set(np.unique(S)) <= {0, 1} and 0.4 < S.mean() < 0.6
    msgs = np.arange(64, dtype=np.uint64) % 16
    phases = msgs << np.uint64(59)
    cts = keys.encrypt(phases)
    # device decrypt == oracle decrypt (integer, bit-exact) and both close to the plaintext phases
    ph_dev = keys.decrypt(cts)
    ph_ref = oracle.lwe_phase(S, D_SMALL, cts)
    assert np.array_equal(ph_dev, ph_ref)
    assert np.abs(_centered(ph_dev - phases)).max() < 2.0 ** -40


def test_keyswitch_bit_exact(keys, oracle):
    S, s = keys.export_secret()
    rng = np.random.default_rng(0)
    phases = rng.integers(0, 16, 50).astype(np.uint64) << np.uint64(59)
    cts = keys.encrypt(phases)
    for tier in (0, 1):
        t = TIERS_SMALL[tier]
        ksk = keys.export_ksk(tier)
        for shift in (0, 3):
            dev = keys.keyswitch(tier, cts, shift=shift)
            ref = oracle.keyswitch(cts << np.uint64(shift), ksk, t["betak"])
            assert np.array_equal(dev, ref), (tier, shift)
        # and the key itself is a valid key-switch key: message survives
        ph = oracle.lwe_phase(s[: t["n"]].copy(), t["n"], keys.keyswitch(tier, cts))
        assert np.abs(_centered(ph - phases)).max() < 2.0 ** -6


def test_keyswitch_on_effective_dimension(keys, oracle):
    """Ciphertexts with a zero tail (nested keys: outputs of a smaller ring): switching over the first deff key rows only
    gives the full key switch bit for bit -- device against itself and against the oracle."""
    rng = np.random.default_rng(4)
    for tier, deff in ((0, 512), (1, 256), (0, 1024)):
        t = TIERS_SMALL[tier]
        cts = rng.integers(0, 2 ** 64, (37, D_SMALL + 1), dtype=np.uint64)
        cts[:, deff:D_SMALL] = 0
        full = keys.keyswitch(tier, cts, shift=2)
        pref = keys.keyswitch(tier, cts, shift=2, deff=deff)
        assert np.array_equal(full, pref), (tier, deff)
        assert np.array_equal(pref, oracle.keyswitch(cts << np.uint64(2), keys.export_ksk(tier), t["betak"]))


@pytest.mark.parametrize("tier,w", [(0, 4), (1, 3), (2, 4)])
def test_pbs_all_messages(keys, oracle, tier, w):
    """decrypt(PBS_f(enc(m))) == f(m) for every m; device and oracle agree on decrypted values."""
    S, s = keys.export_secret()
    t = TIERS_SMALL[tier]
    N = 1 << t["logN"]
    msgs = np.arange(1 << w, dtype=np.uint64)
    phases = msgs << np.uint64(63 - w)
    small = oracle.lwe_encrypt(s[: t["n"]].copy(), t["n"], phases, 2.0 ** -30, seed=11)
    f = (msgs * 5 + 3) % (1 << w)
    table = (f.astype(np.int64)) << (63 - w - 2)          # output with two spare bits
    dev = keys.pbs(tier, small, table, w)
    bsk = keys.export_bsk(tier)
    bskf = oracle.bsk_to_fourier(bsk)
    ref = oracle.pbs(small, bskf, bsk, t["k"], N, t["l"], t["beta"], table, w, None, D_SMALL)
    ph_dev = oracle.lwe_phase(S, D_SMALL, dev)
    ph_ref = oracle.lwe_phase(S, D_SMALL, ref)
    want = table.astype(np.uint64)
    err_dev, err_ref = np.abs(_centered(ph_dev - want)), np.abs(_centered(ph_ref - want))
    dec = lambda ph: ((ph + (np.uint64(1) << np.uint64(63 - w - 3))) >> np.uint64(63 - w - 2)) & np.uint64((1 << (w + 2)) - 1)
    assert np.array_equal(dec(ph_dev), f.astype(np.uint64))
    assert np.array_equal(dec(ph_ref), f.astype(np.uint64))
    # noise of the device result is of the same size as the oracle's (same scheme, same keys)
    assert err_dev.max() < max(4 * err_ref.max(), 2.0 ** -30), (err_dev.max(), err_ref.max())
    # mask beyond k*N stays zero (nested keys)
    assert not dev[:, t["k"] * N: D_SMALL].any()


def test_pbs_two_bit_rotation_k2(gpu_ctx, oracle):
    """the general two-bit rotation on k = 2, N = 1024, one level (tier Ba2: one wave per ciphertext) against its definition"""
    from dctfhe.engine import Keys, make_params
    D, w, N = 2048, 3, 1024
    tier = dict(n=40, k=2, logN=10, l=1, beta=20, lk=4, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -52, unroll=2)
    k = Keys(gpu_ctx, make_params(D, 40, [tier], 2.0 ** -50), seed=23)
    try:
        S, s = k.export_secret()
        msgs = np.arange(1 << w, dtype=np.uint64)
        small = oracle.lwe_encrypt(s[:40].copy(), 40, msgs << np.uint64(63 - w), 2.0 ** -30, seed=14)
        f = (msgs * 5 + 2) % (1 << w)
        table = f.astype(np.int64) << (63 - w - 2)
        dev = k.pbs(0, np.concatenate([small, small, small]), table, w)[: 1 << w]          # 24 ciphertexts: partial workgroups too
        ref = oracle.pbs_mb2(small, k.export_bsk(0), 2, N, 1, 20, table, w, None, D)
        ph_dev, ph_ref = oracle.lwe_phase(S, D, dev), oracle.lwe_phase(S, D, ref)
        dec = lambda ph: ((ph + (np.uint64(1) << np.uint64(63 - w - 3))) >> np.uint64(63 - w - 2)) & np.uint64((1 << (w + 2)) - 1)
        assert np.array_equal(dec(ph_dev), f.astype(np.uint64)) and np.array_equal(dec(ph_ref), f.astype(np.uint64))
        want = table.astype(np.uint64)
        err_dev, err_ref = np.abs(_centered(ph_dev - want)), np.abs(_centered(ph_ref - want))
        assert err_dev.max() < max(4 * err_ref.max(), 2.0 ** -30), (err_dev.max(), err_ref.max())
    finally:
        k.close()


def test_pbs_two_bit_rotation_k2_key_tiles_through_lds(gpu_ctx, oracle):
    """tier.key_lds = 1: the eight ciphertexts of a workgroup share every key tile through an LDS ring filled by LDS-DMA (pbs_core.h,
    KLDS) -- against the same definition, and ciphertext for ciphertext the same decrypted values as the free-running kernel.
    20 ciphertexts: two full workgroups and a padded one."""
    from dctfhe.engine import Keys, make_params
    D, w, N = 2048, 3, 1024
    base = dict(n=40, k=2, logN=10, l=1, beta=20, lk=4, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -52, unroll=2)
    k = Keys(gpu_ctx, make_params(D, 40, [dict(base, key_lds=1), dict(base, ksk_share=0)], 2.0 ** -50), seed=23)
    try:
        S, s = k.export_secret()
        msgs = np.arange(20, dtype=np.uint64) % (1 << w)
        small = oracle.lwe_encrypt(s[:40].copy(), 40, msgs << np.uint64(63 - w), 2.0 ** -30, seed=15)
        f = (np.arange(1 << w, dtype=np.uint64) * 5 + 2) % (1 << w)
        table = f.astype(np.int64) << (63 - w - 2)
        dev = k.pbs(0, small, table, w)
        free = k.pbs(1, small, table, w)
        ref = oracle.pbs_mb2(small[:8], k.export_bsk(0), 2, N, 1, 20, table, w, None, D)
        dec = lambda ph: ((ph + (np.uint64(1) << np.uint64(63 - w - 3))) >> np.uint64(63 - w - 2)) & np.uint64((1 << (w + 2)) - 1)
        ph_dev, ph_free, ph_ref = oracle.lwe_phase(S, D, dev), oracle.lwe_phase(S, D, free), oracle.lwe_phase(S, D, ref)
        assert np.array_equal(dec(ph_dev), f[msgs]) and np.array_equal(dec(ph_free), f[msgs]) and np.array_equal(dec(ph_ref), f[msgs[:8]])
        want = table.astype(np.uint64)[msgs]
        err_dev, err_ref = np.abs(_centered(ph_dev - want)), np.abs(_centered(ph_ref - want[:8]))
        assert err_dev.max() < max(4 * err_ref.max(), 2.0 ** -30), (err_dev.max(), err_ref.max())
        assert not dev[:, 2 * N: D].any()
    finally:
        k.close()


@pytest.mark.parametrize("logN,w,beta", [(13, 6, 22), (12, 5, 22)])
def test_pbs_two_bit_rotation_shipped_big_rings(gpu_ctx, oracle, logN, w, beta):
    """The two instantiations that carry half of an image -- pbs_kernel<13,1,1,8,1,1> (tier T6a: one ciphertext per 512-thread
    workgroup, twist bases read from global memory, 32-part L2 warm-up) and <12,1,1,8,2,1> (T5a: two ciphertexts per workgroup) --
    against the exact-arithmetic definition of the two-bit rotation (oracle ref_pbs_mb2_batch), at the shipped ring, gadget and
    table width with a short key (n = 40: the exact product is ~N^2 per external product)."""
    from dctfhe import params as P
    from dctfhe.engine import Keys, make_params
    D, N, n = 8192, 1 << logN, 40
    tier = dict(n=n, k=1, logN=logN, l=1, beta=beta, lk=4, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -62, unroll=2)
    k = Keys(gpu_ctx, make_params(D, n, [tier], 2.0 ** -62), seed=29)
    try:
        S, s = k.export_secret()
        msgs = np.array([0, 5, 13, 21, 31, 42, 50, 63], dtype=np.uint64) % (1 << w)
        small = oracle.lwe_encrypt(s[:n].copy(), n, msgs << np.uint64(63 - w), 2.0 ** -30, seed=16)
        f = (np.arange(1 << w, dtype=np.uint64) * 7 + 3) % 16                      # 4-bit outputs, as the shipped tables have
        table = f.astype(np.int64) << 57
        dev = k.pbs(0, small, table, w)
        ref = oracle.pbs_mb2(small, k.export_bsk(0), 1, N, 1, beta, table, w, None, D)
        ph_dev, ph_ref = oracle.lwe_phase(S, D, dev), oracle.lwe_phase(S, D, ref)
        dec = lambda ph: ((ph + (np.uint64(1) << np.uint64(56))) >> np.uint64(57)) & np.uint64(63)
        assert np.array_equal(dec(ph_dev), f[msgs]) and np.array_equal(dec(ph_ref), f[msgs])
        want = table.astype(np.uint64)[msgs]
        err_dev, err_ref = np.abs(_centered(ph_dev - want)), np.abs(_centered(ph_ref - want))
        # the definition is exact arithmetic; the device's one-level f64 transform carries the N^2 B^2 error the compiler prices
        model = P.var_pbs_out(P.TierSpec("t", n=n, k=1, logN=logN, l=1, beta=beta, lk=4, betak=4, unroll=2, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -62)) ** 0.5
        assert err_dev.max() < max(4 * err_ref.max(), 5 * model), (err_dev.max(), err_ref.max(), model)
        assert not dev[:, N: D].any()
    finally:
        k.close()


@pytest.mark.parametrize("l,beta", [(1, 20), (3, 12)])
def test_pbs_two_bit_rotation(gpu_ctx, oracle, l, beta):
    """tier.unroll == 2 (two key bits per blind-rotate iteration, csrc/pbs_core.h): every message decodes to f(m), the
    exported key is a bootstrapping key of the pair secret, and the device agrees with the exact-arithmetic definition
    (oracle ref_pbs_mb2_batch) on decrypted values and on the size of the noise.  l = 1: the paired kernel of the one-level
    tiers; l = 3: the general form (tier T4r2: one polynomial at a time, four ciphertexts per workgroup)."""
    from dctfhe.engine import Keys, make_params
    D, w = 2048, 3
    tier = dict(n=40, k=1, logN=11, l=l, beta=beta, lk=4, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -52, unroll=2)
    k = Keys(gpu_ctx, make_params(D, 40, [tier], 2.0 ** -50), seed=21)
    try:
        S, s = k.export_secret()
        msgs = np.arange(1 << w, dtype=np.uint64)
        small = oracle.lwe_encrypt(s[:40].copy(), 40, msgs << np.uint64(63 - w), 2.0 ** -30, seed=13)
        f = (msgs * 3 + 1) % (1 << w)
        table = f.astype(np.int64) << (63 - w - 2)
        dev = k.pbs(0, small, table, w)
        bsk3 = k.export_bsk(0)
        assert bsk3.shape[0] == 60                                   # 3n/2 key blocks
        # block 3i+v of the key encrypts bit v of the pair secret: phase of the body row's gadget coefficient
        ps = oracle.pair_secret(s[:40].copy())
        assert ps.reshape(-1, 3).sum(axis=1).max() <= 1 and ps.sum() > 0
        ref = oracle.pbs_mb2(small, bsk3, 1, 2048, l, beta, table, w, None, D)
        ph_dev, ph_ref = oracle.lwe_phase(S, D, dev), oracle.lwe_phase(S, D, ref)
        want = table.astype(np.uint64)
        dec = lambda ph: ((ph + (np.uint64(1) << np.uint64(63 - w - 3))) >> np.uint64(63 - w - 2)) & np.uint64((1 << (w + 2)) - 1)
        assert np.array_equal(dec(ph_dev), f.astype(np.uint64))
        assert np.array_equal(dec(ph_ref), f.astype(np.uint64))
        err_dev, err_ref = np.abs(_centered(ph_dev - want)), np.abs(_centered(ph_ref - want))
        # the exact-arithmetic definition carries no transform error; the device's f64 FFT does (params.var_pbs_out prices it:
        # with three levels of 12-bit digits it dominates, sigma ~2^-25 at this n)
        from dctfhe import params as P
        model = P.var_pbs_out(P.TierSpec("t", n=40, k=1, logN=11, l=l, beta=beta, lk=4, betak=4, unroll=2, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -52)) ** 0.5
        assert err_dev.max() < max(4 * err_ref.max(), 2.0 ** -30, 4 * model), (err_dev.max(), err_ref.max(), model)
    finally:
        k.close()


def test_pbs_negacyclic_rule(keys, oracle):
    """LUT[m + 2^w] = -LUT[m]: a message with the padding bit set comes back negated."""
    S, s = keys.export_secret()
    t, w = TIERS_SMALL[0], 3
    msgs = np.arange(1 << (w + 1), dtype=np.uint64)
    small = oracle.lwe_encrypt(s[: t["n"]].copy(), t["n"], msgs << np.uint64(63 - w), 2.0 ** -30, seed=12)
    table = (np.arange(1, (1 << w) + 1, dtype=np.int64)) << 56
    ph = oracle.lwe_phase(S, D_SMALL, keys.pbs(0, small, table, w))
    got = np.round(_centered(ph) * 2.0 ** 8).astype(np.int64)
    want = np.concatenate([np.arange(1, 9), -np.arange(1, 9)])
    assert np.array_equal(got, want)


def test_round_lut_exact_rounding(keys, oracle):
    """bit-extract rounding == ((m + 2^(r-1)) >> r) for all m, then per-channel tables."""
    S, s = keys.export_secret()
    p, r, w = 7, 3, 4
    msgs = np.arange(0, (1 << p) - (1 << (r - 1)), dtype=np.uint64)      # stay inside the padded range after rounding
    cts = keys.encrypt(msgs << np.uint64(63 - p))
    tables = np.stack([(np.arange(16) * 3 + 1) % 16, (15 - np.arange(16))]).astype(np.int64) << 58
    idx = (np.arange(msgs.size) % 2).astype(np.int32)
    out = keys.round_lut(1, 0, cts, p, r, tables, w, idx)
    ph = keys.decrypt(out)
    got = np.round(_centered(ph) * 2.0 ** 6).astype(np.int64) % 64
    t_idx = ((msgs + np.uint64(1 << (r - 1))) >> np.uint64(r)).astype(np.int64)
    want = (tables >> 58)[idx, t_idx]
    assert np.array_equal(got, want)
    # the oracle's own chain gives the same decrypted values from the same keys
    bt, tt = TIERS_SMALL[1], TIERS_SMALL[0]
    mk = lambda i, t: oracle.make_tier(t["n"], t["k"], 1 << t["logN"], t["l"], t["beta"], t["lk"], t["betak"],
                                       oracle.bsk_to_fourier(keys.export_bsk(i)), keys.export_ksk(i))
    ref = oracle.round_lut(cts[:16], D_SMALL, p, r, mk(1, bt), mk(0, tt), tables, w, idx[:16])
    got_ref = np.round(_centered(oracle.lwe_phase(S, D_SMALL, ref)) * 2.0 ** 6).astype(np.int64) % 64
    assert np.array_equal(got_ref, want[:16])


def test_conv2d_bit_exact(gpu_ctx, oracle):
    rng = np.random.default_rng(1)
    D, B, Cin, H, W, Cout = 96, 2, 5, 6, 7, 19
    cts = rng.integers(0, 2 ** 64, (B, Cin, H, W, D + 1), dtype=np.uint64)
    for (K, stride, pad) in [(3, 1, 1), (1, 1, 0), (3, 2, 1), (1, 2, 0)]:
        wgt = rng.integers(-7, 8, (Cout, Cin, K, K)).astype(np.int8)
        dev = gpu_ctx.conv2d(D, cts, B, Cin, H, W, wgt, stride, pad)
        for b in range(B):
            ref = oracle.conv2d(cts[b], Cin, H, W, D, wgt.astype(np.int32), stride, pad)
            assert np.array_equal(dev[b], ref), (K, stride, pad, b)


def test_fp64_peak_probe(gpu_ctx):
    tf = gpu_ctx.fp64_peak()
    print("measured f64 FMA peak: %.1f TFLOP/s" % tf)
    assert 20 < tf < 200


def test_keyswitch_valu_fallback_bit_exact(gpu_ctx, oracle):
    """D*lk not a multiple of 64: the engine takes the integer-VALU key-switch GEMM instead of the MFMA one."""
    from dctfhe.engine import Keys, make_params
    D = 1040
    tier = dict(n=33, k=1, logN=9, l=2, beta=12, lk=3, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -45)
    k = Keys(gpu_ctx, make_params(D, 33, [tier], 2.0 ** -50), seed=9)
    try:
        rng = np.random.default_rng(3)
        cts = k.encrypt(rng.integers(0, 16, 37).astype(np.uint64) << np.uint64(59))
        assert np.array_equal(k.keyswitch(0, cts, shift=2), oracle.keyswitch(cts << np.uint64(2), k.export_ksk(0), 4))
    finally:
        k.close()


@pytest.mark.parametrize("betak,lk", [(7, 4), (8, 3)])
def test_keyswitch_widest_gadgets_bit_exact(gpu_ctx, oracle, betak, lk):
    """the i8 MFMA key switch reads its digit operand as SIGNED bytes, so offset digits in [0, 2^betak) need
    betak <= 7 (the widest it takes, checked here on a shape the MFMA tiling covers); betak = 8 is accepted by
    dctfhe_params_check and must go to the integer-VALU GEMM -- bit-exact either way."""
    from dctfhe.engine import Keys, make_params
    D = 1024
    tier = dict(n=40, k=1, logN=10, l=2, beta=10, lk=lk, betak=betak, lwe_sigma=2.0 ** -40, glwe_sigma=2.0 ** -45)
    k = Keys(gpu_ctx, make_params(D, 40, [tier], 2.0 ** -50), seed=31)
    try:
        rng = np.random.default_rng(5)
        cts = k.encrypt(rng.integers(0, 16, 150).astype(np.uint64) << np.uint64(59))
        cts[:, :D] |= rng.integers(0, 2 ** 64, (150, D), dtype=np.uint64) & np.uint64(0xFF00000000000000)     # every digit value occurs, the top ones too
        assert np.array_equal(k.keyswitch(0, cts), oracle.keyswitch(cts, k.export_ksk(0), betak))
    finally:
        k.close()
