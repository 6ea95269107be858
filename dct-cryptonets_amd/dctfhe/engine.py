"""Object layer over the C ABI: Context, Keys, Circuit, Session (host numpy buffers in and out)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Params, Stats, Tier, Timing, check, ptr


def make_params(D, n_max, tiers, input_sigma, input_dim=0):
    """tiers: list of dicts with n,k,logN,l,beta,lk,betak,lwe_sigma,glwe_sigma[,ksk_share][,unroll]."""
    p = Params()
    p.D, p.n_max, p.n_tiers, p.input_sigma, p.input_dim = D, n_max, len(tiers), input_sigma, input_dim
    for i, t in enumerate(tiers):
        p.tiers[i] = Tier(t["n"], t["k"], t["logN"], t["l"], t["beta"], t["lk"], t["betak"], t.get("ksk_share", -1),
                          t.get("unroll", 1), 0, t["lwe_sigma"], t["glwe_sigma"])
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


class Keys:
    def __init__(self, ctx, params, seed):
        self.ctx, self.params, self.L = ctx, params, ctx.L
        self.h = C.c_void_p()
        check(self.L.dctfhe_keygen(ctx.h, C.byref(params), seed, C.byref(self.h)))

    @property
    def D(self):
        return self.params.D

    def tier(self, i):
        return self.params.tiers[i]

    def export_secret(self):
        S = np.empty(self.params.D, np.uint8)
        s = np.empty(self.params.n_max, np.uint8)
        check(self.L.dctfhe_keys_export_secret(self.h, ptr(S), ptr(s)))
        return S, s

    def export_ksk(self, tier):
        t = self.tier(tier)
        out = np.empty((self.params.D, t.lk, t.n + 1), np.uint64)
        check(self.L.dctfhe_keys_export_ksk(self.h, tier, ptr(out)))
        return out

    def export_bsk(self, tier):
        t = self.tier(tier)
        blocks = 3 * t.n // 2 if t.unroll == 2 else t.n        # unroll 2: the key of the pair secret
        out = np.empty((blocks, (t.k + 1) * t.l, t.k + 1, 1 << t.logN), np.uint64)
        check(self.L.dctfhe_keys_export_bsk(self.h, tier, ptr(out)))
        return out

    def encrypt(self, phases, seed):
        phases = np.ascontiguousarray(phases, np.uint64).reshape(-1)
        out = np.empty((phases.size, self.D + 1), np.uint64)
        check(self.L.dctfhe_encrypt(self.ctx.h, self.h, ptr(phases), phases.size, seed, ptr(out)))
        return out

    def decrypt(self, cts):
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, self.D + 1)
        out = np.empty(cts.shape[0], np.uint64)
        check(self.L.dctfhe_decrypt(self.ctx.h, self.h, ptr(cts), cts.shape[0], ptr(out)))
        return out

    def keyswitch(self, tier, cts, shift=0, deff=0):
        """deff > 0: the caller knows every mask word beyond deff to be zero (see dctfhe_keyswitch_prefix)"""
        cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, self.D + 1)
        out = np.empty((cts.shape[0], self.tier(tier).n + 1), np.uint64)
        check(self.L.dctfhe_keyswitch_prefix(self.ctx.h, self.h, tier, ptr(cts), cts.shape[0], shift, deff, ptr(out)))
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
            self.L.dctfhe_keys_destroy(self.h)
            self.h = C.c_void_p()


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
        self.ctx, self.circuit, self.keys, self.batch, self.L = ctx, circuit, keys, batch, ctx.L
        self.words = (keys.D + 1) if keys is not None else 1
        self.h = C.c_void_p()
        check(self.L.dctfhe_session_create(ctx.h, circuit.h, keys.h if keys is not None else None, batch, C.byref(self.h)))

    def upload(self, cts):
        cts = np.ascontiguousarray(cts, np.uint64)
        assert cts.size == self.batch * self.circuit.n_in * self.words, (cts.shape, self.batch, self.circuit.n_in, self.words)
        check(self.L.dctfhe_session_upload(self.h, ptr(cts)))

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

    def download(self):
        out = np.empty((self.batch, self.circuit.n_out, self.words), np.uint64)
        check(self.L.dctfhe_session_download(self.h, ptr(out)))
        return out

    def close(self):
        if self.h:
            self.L.dctfhe_session_destroy(self.h)
            self.h = C.c_void_p()
