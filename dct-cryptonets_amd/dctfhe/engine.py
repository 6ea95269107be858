"""Object layer over the C ABI: Context, Keys, Circuit, Session (host numpy buffers in and out)."""
import ctypes as C
import hashlib
import os

import numpy as np

from . import _lib
from ._lib import Params, Stats, Tier, Timing, check, ptr


def make_params(D, n_max, tiers, input_sigma, input_dim=0):
    """tiers: list of dicts with n,k,logN,l,beta,lk,betak,lwe_sigma,glwe_sigma[,ksk_share][,unroll][,key_lds]."""
    p = Params()
    p.D, p.n_max, p.n_tiers, p.input_sigma, p.input_dim = D, n_max, len(tiers), input_sigma, input_dim
    for i, t in enumerate(tiers):
        p.tiers[i] = Tier(t["n"], t["k"], t["logN"], t["l"], t["beta"], t["lk"], t["betak"], t.get("ksk_share", -1),
                          t.get("unroll", 1), t.get("key_lds", 0), t["lwe_sigma"], t["glwe_sigma"])
    return p


class Context:
    def __init__(self, device=0):
        self.L = _lib.load()
        self.h = C.c_void_p()
        check(self.L.dctfhe_ctx_create(device, C.byref(self.h)))

    def set_stream(self, stream_ptr):
        check(self.L.dctfhe_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        check(self.L.dctfhe_ctx_synchronize(self.h))

    def fp64_peak(self):
        v = C.c_double()
        check(self.L.dctfhe_fp64_peak(self.h, C.byref(v)))
        return v.value

    def dct_frontend(self, y, c1, c2, fs, idx, mean, std, round_coeffs=False):
        """uint8 planes y [B, fs*S, fs*S], c1/c2 [B, fs*Sc, fs*Sc] -> float32 [B, C, S, S] (include/dctfhe.h dctfhe_dct_frontend)"""
        y, c1, c2 = (np.ascontiguousarray(a, np.uint8) for a in (y, c1, c2))
        B, S, Sc = y.shape[0], y.shape[1] // fs, c1.shape[1] // fs
        ii = [np.ascontiguousarray(i, np.int32) for i in idx]
        mean, std = np.ascontiguousarray(mean, np.float32), np.ascontiguousarray(std, np.float32)
        C_ = sum(i.size for i in ii)
        assert mean.size == std.size == C_ and y.shape == (B, fs * S, fs * S) and c1.shape == c2.shape == (B, fs * Sc, fs * Sc)
        out = np.empty((B, C_, S, S), np.float32)
        check(self.L.dctfhe_dct_frontend(self.h, ptr(y), ptr(c1), ptr(c2), B, S, Sc, fs, ptr(ii[0]), ii[0].size, ptr(ii[1]), ii[1].size,
                                         ptr(ii[2]), ii[2].size, ptr(mean), ptr(std), int(bool(round_coeffs)), ptr(out)))
        return out

    def add_rows(self, a, deff_a, b, deff_b, dim_o):
        """rows [count, dim + 1] at effective dimensions deff_* -> a + b as rows [count, dim_o + 1] (include/dctfhe.h dctfhe_add_rows)"""
        a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
        out = np.empty((a.shape[0], dim_o + 1), np.uint64)
        check(self.L.dctfhe_add_rows(self.h, ptr(a), a.shape[1] - 1, deff_a, ptr(b), b.shape[1] - 1, deff_b, a.shape[0], dim_o, ptr(out)))
        return out

    def affine_rows(self, a, deff_a, inout, nwords, shift, body_add):
        a = np.ascontiguousarray(a, np.uint64)
        out = np.ascontiguousarray(inout, np.uint64).copy()
        check(self.L.dctfhe_affine_rows(self.h, ptr(a), a.shape[1] - 1, deff_a, a.shape[0], nwords, shift, C.c_uint64(body_add), out.shape[1] - 1, ptr(out)))
        return out

    def sum_pool_rows(self, x, deff, K, dim_o):
        """x [batch, C, H, W, dim + 1] -> [batch, C, H // K, W // K, dim_o + 1]"""
        x = np.ascontiguousarray(x, np.uint64)
        B, Cc, H, W, L = x.shape
        out = np.empty((B, Cc, H // K, W // K, dim_o + 1), np.uint64)
        check(self.L.dctfhe_sum_pool_rows(self.h, ptr(x), L - 1, deff, B, Cc, H, W, K, dim_o, ptr(out)))
        return out

    def conv2d(self, D, cts, batch, Cin, H, W, weight, stride, pad):
        weight = np.ascontiguousarray(weight, np.int8)
        Cout, _, KH, KW = weight.shape
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        cts = np.ascontiguousarray(cts, np.uint64)
        out = np.empty((batch, Cout, Ho, Wo, D + 1), np.uint64)
        check(self.L.dctfhe_conv2d(self.h, D, ptr(cts), batch, Cin, H, W, ptr(weight), Cout, KH, KW, stride, pad, ptr(out)))
        return out

    def close(self):
        if self.h:
            self.L.dctfhe_ctx_destroy(self.h)
            self.h = C.c_void_p()


def seed_bytes(seed=None):
    """The 32-byte CSPRNG seed of a client key.  None: fresh from the OS (os.urandom) -- the default everywhere;
    32 bytes: used as they are (a persisted or broadcast key); int: a DETERMINISTIC seed for tests and reproducible
    experiments (SHA-256 of a label and the integer) -- guessable by construction, never for real data."""
    if seed is None:
        return os.urandom(32)
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) != 32:
            raise ValueError("a key seed is exactly 32 bytes")
        return bytes(seed)
    return hashlib.sha256(b"dctfhe deterministic test seed " + str(int(seed)).encode()).digest()


class ClientKey:
    """Secret side (include/dctfhe.h dctfhe_client_key): encrypts, decrypts, generates evaluation keys."""

    def __init__(self, ctx, params, seed=None):
        self.ctx, self.params, self.L = ctx, params, ctx.L
        self.seed = seed_bytes(seed)
        self.h = C.c_void_p()
        check(self.L.dctfhe_client_key_create(ctx.h, C.byref(params), self.seed, C.byref(self.h)))

    @property
    def D(self):
        return self.params.D

    def tier(self, i):
        return self.params.tiers[i]

    def generate_eval_keys(self):
        h = C.c_void_p()
        check(self.L.dctfhe_eval_keys_generate(self.h, C.byref(h)))
        return EvalKeys(self.ctx, self.params, h)

    def export_secret(self):
        S = np.empty(self.params.D, np.uint8)
        s = np.empty(self.params.n_max, np.uint8)
        check(self.L.dctfhe_client_key_export_secret(self.h, ptr(S), ptr(s)))
        return S, s

    def export_bsk(self, tier):
        t = self.tier(tier)
        blocks = 3 * t.n // 2 if t.unroll == 2 else t.n        # unroll 2: the key of the pair secret
        out = np.empty((blocks, (t.k + 1) * t.l, t.k + 1, 1 << t.logN), np.uint64)
        check(self.L.dctfhe_client_key_export_bsk(self.h, tier, ptr(out)))
        return out

    def set_encrypt_counter(self, next_call):
        """position inside this handle's own encryption streams (handles already differ by their nonce: include/dctfhe.h)"""
        check(self.L.dctfhe_client_key_set_encrypt_counter(self.h, next_call))

    def set_encrypt_nonce(self, nonce16):
        """FIX the 128-bit encryption nonce the handle drew from the OS -- reproducible tests / experiments only"""
        nonce16 = bytes(nonce16)
        if len(nonce16) != 16:
            raise ValueError("an encryption nonce is exactly 16 bytes")
        check(self.L.dctfhe_client_key_set_encrypt_nonce(self.h, nonce16))

    @property
    def input_dim(self):
        """mask words of a fresh encryption (the compact row width of a circuit input)"""
        return self.params.input_dim or self.params.D

    def encrypt(self, phases, dim=None):
        """-> rows of dim mask words + body; dim None: the full-width form (D + 1 words), dim = self.input_dim: the compact wire form"""
        phases = np.ascontiguousarray(phases, np.uint64).reshape(-1)
        dim = self.D if dim is None else int(dim)
        out = np.empty((phases.size, dim + 1), np.uint64)
        check(self.L.dctfhe_encrypt_rows(self.ctx.h, self.h, ptr(phases), phases.size, dim, ptr(out)))
        return out

    def decrypt(self, cts, dim=None):
        dim = self.D if dim is None else int(dim)
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, dim + 1)
        out = np.empty(cts.shape[0], np.uint64)
        check(self.L.dctfhe_decrypt_rows(self.ctx.h, self.h, ptr(cts), cts.shape[0], dim, ptr(out)))
        return out

    def close(self):
        if self.h:
            self.L.dctfhe_client_key_destroy(self.h)
            self.h = C.c_void_p()


class EvalKeys:
    """Server side (dctfhe_eval_keys): key-switch keys + Fourier bootstrap keys; all the evaluation needs."""

    def __init__(self, ctx, params, handle):
        self.ctx, self.params, self.L, self.h = ctx, params, ctx.L, handle

    @classmethod
    def from_blob(cls, ctx, blob):
        """evaluation keys as shipped by a client (EvalKeys.to_blob): the server never sees a secret"""
        blob = np.ascontiguousarray(np.frombuffer(blob, np.uint8) if isinstance(blob, (bytes, bytearray, memoryview)) else blob, np.uint8)
        h = C.c_void_p()
        check(ctx.L.dctfhe_eval_keys_import(ctx.h, ptr(blob), blob.size, C.byref(h)))
        params = Params.from_buffer_copy(blob[16:16 + C.sizeof(Params)].tobytes())       # header: magic, version, total_bytes, params
        return cls(ctx, params, h)

    def to_blob(self):
        n = C.c_size_t()
        check(self.L.dctfhe_eval_keys_export(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint8)
        check(self.L.dctfhe_eval_keys_export(self.h, ptr(out), out.size, C.byref(n)))
        return out

    @property
    def D(self):
        return self.params.D

    def tier(self, i):
        return self.params.tiers[i]

    def export_ksk(self, tier):
        t = self.tier(tier)
        out = np.empty((self.params.D, t.lk, t.n + 1), np.uint64)
        check(self.L.dctfhe_eval_keys_export_ksk(self.h, tier, ptr(out)))
        return out

    def keyswitch(self, tier, cts, shift=0, deff=0):
        """deff > 0: the caller knows every mask word beyond deff to be zero (see dctfhe_keyswitch_prefix)"""
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, self.D + 1)
        out = np.empty((cts.shape[0], self.tier(tier).n + 1), np.uint64)
        check(self.L.dctfhe_keyswitch_prefix(self.ctx.h, self.h, tier, ptr(cts), cts.shape[0], shift, deff, ptr(out)))
        return out

    def modswitch_center(self, tier, cts_small):
        """centred mod switch (dctfhe_modswitch_center): the adjusted copy of small ciphertexts [count, n + 1]"""
        out = np.ascontiguousarray(cts_small, np.uint64).copy()
        check(self.L.dctfhe_modswitch_center(self.ctx.h, self.h, tier, ptr(out), out.shape[0]))
        return out

    def pbs(self, tier, cts_small, tables, w, table_idx=None):
        cts_small = np.ascontiguousarray(cts_small, np.uint64)
        tables = np.ascontiguousarray(tables, np.int64).reshape(-1, 1 << w)
        idx = None if table_idx is None else np.ascontiguousarray(table_idx, np.int32)
        out = np.empty((cts_small.shape[0], self.D + 1), np.uint64)
        check(self.L.dctfhe_pbs(self.ctx.h, self.h, tier, ptr(cts_small), cts_small.shape[0], ptr(tables), tables.shape[0], w,
                                None if idx is None else ptr(idx), ptr(out)))
        return out

    def round_lut(self, bit_tier, tab_tier, cts, p, r, tables, w, table_idx=None):
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, self.D + 1)
        tables = np.ascontiguousarray(tables, np.int64).reshape(-1, 1 << w)
        idx = None if table_idx is None else np.ascontiguousarray(table_idx, np.int32)
        out = np.empty_like(cts)
        check(self.L.dctfhe_round_lut(self.ctx.h, self.h, bit_tier, tab_tier, ptr(cts), cts.shape[0], p, r, ptr(tables),
                                      tables.shape[0], w, None if idx is None else ptr(idx), ptr(out)))
        return out

    def bench_pbs(self, tier, count, reps=3):
        v = C.c_double()
        check(self.L.dctfhe_bench_pbs(self.ctx.h, self.h, tier, count, reps, C.byref(v)))
        return v.value

    def close(self):
        if self.h:
            self.L.dctfhe_eval_keys_destroy(self.h)
            self.h = C.c_void_p()


class Keys:
    """Both halves in one process (tests, benchmarks, the reference's single-machine flow homomorphic_eval.py:313-317):
    `.client` (ClientKey) and `.eval` (EvalKeys); every method is the half's own."""

    def __init__(self, ctx, params, seed=None, client=None, evalk=None):
        self.ctx, self.params = ctx, params
        self.client = client if client is not None else ClientKey(ctx, params, seed)
        self.eval = evalk if evalk is not None else self.client.generate_eval_keys()

    @property
    def D(self):
        return self.params.D

    def tier(self, i):
        return self.params.tiers[i]

    def __getattr__(self, name):
        if name in ("export_secret", "export_bsk", "encrypt", "decrypt", "seed", "input_dim", "set_encrypt_nonce", "set_encrypt_counter"):
            return getattr(self.client, name)
        if name in ("export_ksk", "keyswitch", "modswitch_center", "pbs", "round_lut", "bench_pbs", "to_blob"):
            return getattr(self.eval, name)
        raise AttributeError(name)

    def close(self):
        self.eval.close()
        self.client.close()


class Circuit:
    def __init__(self, ctx, blob):
        self.ctx, self.L = ctx, ctx.L
        self.h = C.c_void_p()
        self._blob = bytes(blob)
        check(self.L.dctfhe_circuit_load(ctx.h, self._blob, len(self._blob), C.byref(self.h)))
        a, b = C.c_int64(), C.c_int64()
        check(self.L.dctfhe_circuit_io(self.h, C.byref(a), C.byref(b)))
        self.n_in, self.n_out = a.value, b.value

    def stats(self, params):
        s = Stats()
        check(self.L.dctfhe_circuit_stats(self.h, C.byref(params), C.byref(s)))
        return s

    def close(self):
        if self.h:
            self.L.dctfhe_circuit_destroy(self.h)
            self.h = C.c_void_p()


class Session:
    """Device tensors for one (circuit, keys, batch).  keys=None: noise-free clear mode (1-word ciphertexts)."""

    def __init__(self, ctx, circuit, keys, batch):
        """keys: EvalKeys (or a Keys pair, whose evaluation half is used) -- a session never needs the secret"""
        keys = getattr(keys, "eval", keys)
        self.ctx, self.circuit, self.keys, self.batch, self.L = ctx, circuit, keys, batch, ctx.L
        self.words = (keys.D + 1) if keys is not None else 1
        self.h = C.c_void_p()
        check(self.L.dctfhe_session_create(ctx.h, circuit.h, keys.h if keys is not None else None, batch, C.byref(self.h)))

    def dims(self):
        """(input, output) effective dimensions: the compact row widths of this session's circuit (0, 0 in clear mode)"""
        a, b = C.c_int(), C.c_int()
        check(self.L.dctfhe_session_dims(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def upload(self, cts, dim=None):
        """dim None: rows of D + 1 words (1 in clear mode); else the compact wire form, rows of dim mask words + body"""
        cts = np.ascontiguousarray(cts, np.uint64)
        words = self.words if (dim is None or self.keys is None) else dim + 1
        assert cts.size == self.batch * self.circuit.n_in * words, (cts.shape, self.batch, self.circuit.n_in, words)
        if dim is None or self.keys is None:
            check(self.L.dctfhe_session_upload(self.h, ptr(cts)))
        else:
            check(self.L.dctfhe_session_upload_rows(self.h, ptr(cts), int(dim)))

    def set_noise(self, seed, sigma_per_op):
        """clear-mode sessions: `simulate` with the noise model (sigma per op, fraction of the torus); None switches it off"""
        if sigma_per_op is None:
            check(self.L.dctfhe_session_set_noise(self.h, 0, None, 0))
            return
        sg = np.ascontiguousarray(sigma_per_op, np.float64)
        check(self.L.dctfhe_session_set_noise(self.h, seed, ptr(sg), sg.size))

    def run(self, timing=False):
        t = Timing() if timing else None
        check(self.L.dctfhe_session_run(self.h, C.byref(t) if timing else None))
        return t

    def download(self, dim=None):
        if dim is None or self.keys is None:
            out = np.empty((self.batch, self.circuit.n_out, self.words), np.uint64)
            check(self.L.dctfhe_session_download(self.h, ptr(out)))
        else:
            out = np.empty((self.batch, self.circuit.n_out, dim + 1), np.uint64)
            check(self.L.dctfhe_session_download_rows(self.h, ptr(out), int(dim)))
        return out

    def close(self):
        if self.h:
            self.L.dctfhe_session_destroy(self.h)
            self.h = C.c_void_p()
