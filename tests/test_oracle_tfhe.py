"""Pins the CPU oracle by scheme identities (there are no reference vectors: PARITY UNPINNED, oracle/tfhe_ref.h):
P2 of SURVEY 8(c) on tiny rings."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def kit(oracle):
    D, n, k, N, l, beta, lk, betak = 512, 24, 1, 256, 2, 10, 4, 4
    S, s = oracle.gen_binary_key(1, D), oracle.gen_binary_key(2, n)
    ksk = oracle.ksk_gen(S, s, lk, betak, 2.0 ** -30, 3)
    bsk = oracle.bsk_gen(s, S, k, N, l, beta, 2.0 ** -40, 4)
    return dict(D=D, n=n, k=k, N=N, l=l, beta=beta, lk=lk, betak=betak, S=S, s=s, ksk=ksk, bsk=bsk, bskf=oracle.bsk_to_fourier(bsk))


def _cent(x):
    return x.astype(np.int64).astype(np.float64) / 2.0 ** 64


def test_decompose_recomposes(oracle):
    import ctypes
    rng = np.random.default_rng(0)
    for (l, beta) in [(2, 10), (3, 7), (1, 23), (6, 3)]:
        for v in rng.integers(0, 2 ** 64, 200, dtype=np.uint64):
            d = np.zeros(l, np.int32)
            oracle.lib().ref_decompose(ctypes.c_uint64(int(v)), l, beta, d)
            assert d.min() >= -(1 << (beta - 1)) and d.max() < (1 << (beta - 1))
            rec = sum(int(d[i]) << (64 - beta * (i + 1)) for i in range(l)) % (1 << 64)
            err = (rec - int(v) + (1 << 63)) % (1 << 64) - (1 << 63)
            assert abs(err) <= 1 << (63 - l * beta)


def test_linearity_and_keyswitch(kit, oracle):
    p = 4
    m1, m2 = np.arange(8, dtype=np.uint64), np.arange(8, dtype=np.uint64)[::-1].copy()
    enc = lambda m, seed: oracle.lwe_encrypt(kit["S"], kit["D"], m << np.uint64(63 - p), 2.0 ** -40, seed)
    c1, c2 = enc(m1, 5), enc(m2, 6)
    dec = lambda ph: ((ph + (np.uint64(1) << np.uint64(62 - p))) >> np.uint64(63 - p)) & np.uint64(2 ** (p + 1) - 1)
    assert np.array_equal(dec(oracle.lwe_phase(kit["S"], kit["D"], c1 + c2)), m1 + m2)            # decrypt(c1+c2) == m1+m2
    assert np.array_equal(dec(oracle.lwe_phase(kit["S"], kit["D"], c1 * np.uint64(3))), (3 * m1) % 32)  # decrypt(w*c) == w*m
    small = oracle.keyswitch(c1, kit["ksk"], kit["betak"])
    assert np.array_equal(dec(oracle.lwe_phase(kit["s"], kit["n"], small)), m1)                  # key switch preserves m


@pytest.mark.parametrize("exact", [False, True])
def test_pbs_every_message(kit, oracle, exact):
    w = 4
    msgs = np.arange(16, dtype=np.uint64)
    small = oracle.lwe_encrypt(kit["s"], kit["n"], msgs << np.uint64(63 - w), 2.0 ** -30, 9)
    f = (msgs * 7 + 3) % 16
    table = f.astype(np.int64) << 58
    out = oracle.pbs(small, kit["bskf"], kit["bsk"], kit["k"], kit["N"], kit["l"], kit["beta"], table, w, None, kit["D"], exact=exact)
    ph = oracle.lwe_phase(kit["S"], kit["D"], out)
    assert np.array_equal(np.round(_cent(ph) * 64).astype(np.int64) % 64, f.astype(np.int64))


def test_fft_path_equals_exact_path(kit, oracle):
    """f64-FFT external product == schoolbook product mod 2^64 up to FFT rounding (compared on phases)"""
    w = 3
    small = oracle.lwe_encrypt(kit["s"], kit["n"], np.arange(8, dtype=np.uint64) << np.uint64(60), 2.0 ** -30, 10)
    table = np.arange(8, dtype=np.int64) << 58
    a = oracle.pbs(small, kit["bskf"], kit["bsk"], kit["k"], kit["N"], kit["l"], kit["beta"], table, w, None, kit["D"], exact=False)
    b = oracle.pbs(small, kit["bskf"], kit["bsk"], kit["k"], kit["N"], kit["l"], kit["beta"], table, w, None, kit["D"], exact=True)
    d = _cent(oracle.lwe_phase(kit["S"], kit["D"], a) - oracle.lwe_phase(kit["S"], kit["D"], b))
    assert np.abs(d).max() < 2.0 ** -12        # both carry the gadget rounding noise of different digit choices


def test_negacyclic_sign_rule_and_sign_table(kit, oracle):
    w = 3
    msgs = np.arange(16, dtype=np.uint64)
    small = oracle.lwe_encrypt(kit["s"], kit["n"], msgs << np.uint64(60), 2.0 ** -30, 11)
    table = np.arange(1, 9, dtype=np.int64) << 56
    ph = oracle.lwe_phase(kit["S"], kit["D"], oracle.pbs(small, kit["bskf"], None, kit["k"], kit["N"], kit["l"], kit["beta"], table, w, None, kit["D"]))
    assert np.array_equal(np.round(_cent(ph) * 256).astype(np.int64), np.concatenate([np.arange(1, 9), -np.arange(1, 9)]))
    # w = 0: +v on (-1/4, 1/4), -v on (1/4, 3/4)
    ph_in = np.array([0, 1 << 61, (1 << 63), (1 << 63) + (1 << 61), (1 << 64) - (1 << 61)], dtype=np.uint64)
    small = oracle.lwe_encrypt(kit["s"], kit["n"], ph_in, 2.0 ** -30, 12)
    v = np.array([1 << 60], np.int64)
    ph = oracle.lwe_phase(kit["S"], kit["D"], oracle.pbs(small, kit["bskf"], None, kit["k"], kit["N"], kit["l"], kit["beta"], v, 0, None, kit["D"]))
    assert np.array_equal(np.round(_cent(ph) * 16).astype(np.int64), [1, 1, -1, -1, 1])


def test_round_lut_is_round_half_up(kit, oracle):
    p, r, w = 6, 2, 4
    bt = oracle.make_tier(kit["n"], kit["k"], kit["N"], kit["l"], kit["beta"], kit["lk"], kit["betak"], kit["bskf"], kit["ksk"])
    msgs = np.arange(0, 62, dtype=np.uint64)
    cts = oracle.lwe_encrypt(kit["S"], kit["D"], msgs << np.uint64(63 - p), 2.0 ** -45, 13)
    table = np.arange(16, dtype=np.int64) << 58
    out = oracle.round_lut(cts, kit["D"], p, r, bt, bt, table, w, None)
    got = np.round(_cent(oracle.lwe_phase(kit["S"], kit["D"], out)) * 64).astype(np.int64)
    assert np.array_equal(got, ((msgs + np.uint64(2)) >> np.uint64(2)).astype(np.int64))
