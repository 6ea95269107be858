"""Host-side pieces of the C ABI that need no GPU: the CSPRNG (ChaCha20 known-answer test), the parameter-set and
circuit-blob validators.  On a GPU box the same negative cases are repeated per entry point in tests/test_gpu_abi_errors.py."""
import ctypes as C
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from dctfhe import _lib
    return _lib.load()


# ------------------------------------------------------------------------------------------ CSPRNG
def _chacha20_block_py(key_words, counter_words):
    """RFC 8439 section 2.3 block function, straight from its pseudo-code (state words 12..15 given by the caller)."""
    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF

    def qr(s, a, b, c, d):
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 16)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 12)
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 8)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 7)
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + list(counter_words)
    s = list(init)
    for _ in range(10):
        qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15)
        qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(s, init)]


RFC8439_KEY = bytes(range(32))
RFC8439_BLOCK = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
                 0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]


def test_chacha20_known_answer(L):
    """RFC 8439 section 2.3.2: key 00..1f, block counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00.  In the library's layout
    (64-bit block counter in words 12-13, 64-bit stream id in words 14-15) that is block 1 | 0x09000000 << 32, stream 0x4a000000."""
    kw = struct.unpack("<8I", RFC8439_KEY)
    assert _chacha20_block_py(kw, [1, 0x09000000, 0x4A000000, 0]) == RFC8439_BLOCK
    block, stream = 1 | (0x09000000 << 32), 0x4A000000
    out = np.zeros(8, np.uint64)
    assert L.dctfhe_rng_host(RFC8439_KEY, stream, block * 8, 8, out.ctypes.data_as(C.c_void_p)) == 0
    want = [RFC8439_BLOCK[2 * i] | (RFC8439_BLOCK[2 * i + 1] << 32) for i in range(8)]
    assert out.tolist() == want


def test_rng_streams_are_counter_based(L):
    """any index in any order; distinct streams / keys give unrelated outputs"""
    key = bytes(range(100, 132))
    a = np.zeros(40, np.uint64)
    L.dctfhe_rng_host(key, 7, 0, 40, a.ctypes.data_as(C.c_void_p))
    b = np.zeros(11, np.uint64)
    L.dctfhe_rng_host(key, 7, 13, 11, b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a[13:24], b)
    c = np.zeros(40, np.uint64)
    L.dctfhe_rng_host(key, 8, 0, 40, c.ctypes.data_as(C.c_void_p))
    assert not np.any(a == c)
    kw = struct.unpack("<8I", key)
    blk = _chacha20_block_py(kw, [2, 0, 7, 0])
    assert int(a[16]) == blk[0] | (blk[1] << 32)


def test_seed_bytes():
    from dctfhe.engine import seed_bytes
    assert len(seed_bytes()) == 32 and seed_bytes() != seed_bytes()            # OS randomness by default
    assert seed_bytes(5) == seed_bytes(5) != seed_bytes(6)                      # ints: deterministic test seeds
    assert seed_bytes(bytes(range(32))) == bytes(range(32))
    with pytest.raises(ValueError):
        seed_bytes(b"short")


# ------------------------------------------------------------------------------------------ parameters
def _params(**over):
    from dctfhe.engine import make_params
    t = dict(n=40, k=1, logN=10, l=2, beta=10, lk=4, betak=4, lwe_sigma=2.0 ** -30, glwe_sigma=2.0 ** -40)
    t.update(over)
    return make_params(1024, 40, [t], 2.0 ** -50)


def _err(L):
    return L.dctfhe_last_error().decode()


def test_params_check_accepts_the_catalogues(L):
    from dctfhe import params as P
    for ps in (P.default_params(), P.params_for_p_error(0.01), P.test_params()):
        assert L.dctfhe_params_check(C.byref(P.to_c_params(ps))) == 0, _err(L)


@pytest.mark.parametrize("over,needle", [
    (dict(n=0), "n out of range"), (dict(n=41), "n out of range"), (dict(k=2, logN=10), "k*N exceeds D"),
    (dict(l=4), "bad bootstrap gadget"), (dict(l=2, beta=17), "bad bootstrap gadget"), (dict(l=1, beta=29), "bad bootstrap gadget"),
    (dict(unroll=3), "unroll must be 1 or 2"), (dict(unroll=2, l=2), "unroll 2 needs"), (dict(betak=9), "bad key-switch gadget"),
    (dict(lk=0), "bad key-switch gadget"), (dict(logN=7), "k or logN out of range"), (dict(ksk_share=0), "ksk_share must name an earlier tier"),
    # the gate in front of an untrusted evaluation-key blob needs BOTH bounds on every field
    (dict(betak=-3), "bad key-switch gadget"), (dict(betak=0), "bad key-switch gadget"), (dict(lk=-1), "bad key-switch gadget"),
    (dict(lk=64, betak=1), "bad key-switch gadget"), (dict(k=0), "k or logN out of range"), (dict(k=3), "k or logN out of range"),
    (dict(logN=14), "k or logN out of range"), (dict(lwe_sigma=-1.0), "noise parameter"), (dict(glwe_sigma=float("nan")), "noise parameter"),
    (dict(key_lds=1), "key_lds"), (dict(key_lds=2, k=2, logN=10, l=1, beta=20, unroll=2), "k*N exceeds D")])
def test_params_check_rejects(L, over, needle):
    assert L.dctfhe_params_check(C.byref(_params(**over))) != 0
    assert needle in _err(L), _err(L)


def test_params_check_rejects_shapes(L):
    p = _params()
    p.n_tiers = 0
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "n_tiers" in _err(L)
    p = _params()
    p.D = 1022
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "multiple of 4" in _err(L)
    p = _params()
    p.input_dim = 2048
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "input_dim" in _err(L)
    assert L.dctfhe_params_check(None) != 0
    # caps: sizes computed from the parameters (key allocations, the blob size an import compares with) stay far inside size_t
    p = _params()
    p.D = 1 << 17
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "65536" in _err(L)
    p = _params()
    p.n_max = 1 << 20
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "n_max" in _err(L)
    p = _params()
    p.n_max = 0
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "n_max" in _err(L)
    p = _params()
    p.input_sigma = -0.5
    assert L.dctfhe_params_check(C.byref(p)) != 0 and "input_sigma" in _err(L)


# ------------------------------------------------------------------------------------------ circuit blobs
@pytest.fixture(scope="module")
def blob():
    from dctfhe import compile as cc, models, params as P
    calib = np.random.default_rng(0).normal(0, 1, (16, 4, 6, 6))
    return cc.compile_model(models.tiny_resnet_q(), calib, param_set=P.test_params()).blob


def _validate(L, b):
    return L.dctfhe_circuit_validate(bytes(b), len(b))


def test_circuit_validate_accepts_compiled_blobs(L, blob):
    assert _validate(L, blob) == 0, _err(L)


def test_circuit_validate_rejects_malformed_blobs(L, blob):
    hdr = struct.Struct("<IIiiiiii")
    magic, ver, nT, nO, tin, tout, mb, _ = hdr.unpack_from(blob, 0)
    rec0 = 32 + 16 * nT

    def patched(off, fmt, *vals):
        b = bytearray(blob)
        struct.pack_into(fmt, b, off, *vals)
        return b
    cases = [
        (blob[:20], "too short"),
        (patched(0, "<I", 0xDEADBEEF), "magic"),
        (patched(4, "<I", 2), "magic/version"),
        (blob[:rec0 + 50], "truncated"),
        (patched(16, "<i", nT + 3), "input/output tensor"),
        (patched(8, "<i", 0), "tensor/op count"),
        (patched(32, "<i", 0), "empty tensor shape"),
        (patched(rec0, "<i", 9), "unknown type"),
        (patched(rec0 + 4, "<i", nT), "tensor id out of range"),
        (patched(rec0 + 16 + 4 * 3, "<i", 0), "bad convolution geometry"),        # op 0 is the stem conv: stride 0
        (patched(rec0 + 16, "<i", 5), "convolution output shape"),               # Cout that does not match the dst tensor
        (patched(rec0 + 96 - 8, "<q", 3), "weight payload"),                     # payload_len
        (patched(rec0 + 96 - 16, "<q", len(blob)), "payload out of range"),      # payload_off past the end
    ]
    for b, needle in cases:
        assert _validate(L, b) != 0, needle
        assert needle in _err(L), (needle, _err(L))
    # a look-up record: find the first one and break its precision fields
    for i in range(nO):
        typ = struct.unpack_from("<i", blob, rec0 + 96 * i)[0]
        if typ == 4:
            off = rec0 + 96 * i + 16
            for b, needle in [(patched(off, "<i", 70), "bad look-up precision"), (patched(off + 8, "<i", 1), "bad look-up precision"),
                              (patched(off + 24, "<i", 3), "tables for")]:
                assert _validate(L, b) != 0 and needle in _err(L), (needle, _err(L))
            break
    else:
        raise AssertionError("no look-up op in the blob")
    assert L.dctfhe_circuit_validate(None, 0) != 0
