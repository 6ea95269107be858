"""ctypes binding of oracle/libtfhe_ref.so (the CPU twin).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  PARITY UNPINNED (see oracle/tfhe_ref.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class RefTier(C.Structure):
    _fields_ = [("n", C.c_int), ("k", C.c_int), ("N", C.c_int), ("l", C.c_int), ("beta", C.c_int),
                ("lk", C.c_int), ("betak", C.c_int), ("bsk_f", C.c_void_p), ("ksk", C.c_void_p)]


def build(force=False):
    so = os.path.join(_HERE, "libtfhe_ref.so")
    src = [os.path.join(_HERE, f) for f in ("tfhe_ref.c", "tfhe_ref.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libtfhe_ref.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    L.ref_gen_binary_key.argtypes = [C.c_uint64, C.c_int, u8p]
    L.ref_lwe_phase_batch.argtypes = [u8p, C.c_int, u64p, C.c_int, u64p]
    L.ref_lwe_encrypt_batch.argtypes = [u8p, C.c_int, C.c_int, u64p, C.c_int, C.c_double, C.c_uint64, u64p]
    L.ref_ksk_gen.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, u64p]
    L.ref_bsk_gen.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, u64p]
    L.ref_bsk_to_fourier.argtypes = [u64p, C.c_int, C.c_int, C.c_int, C.c_int, f64p]
    L.ref_keyswitch.argtypes = [u64p, C.c_int, C.c_int, u64p, C.c_int, C.c_int, C.c_int, u64p]
    L.ref_modswitch.argtypes = [u64p, C.c_int, C.c_int, u32p]
    L.ref_ms_center.argtypes = [u64p, C.c_int, C.c_int, C.c_int]
    L.ref_decompose.argtypes = [C.c_uint64, C.c_int, C.c_int, i32p]
    L.ref_build_testvector.argtypes = [i64p, C.c_int, C.c_int, u64p]
    L.ref_pbs_batch.argtypes = [u64p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, i64p, C.c_int, C.c_void_p, C.c_int, u64p]
    L.ref_pbs_batch.restype = C.c_int
    L.ref_conv2d.argtypes = [u64p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, u64p]
    L.ref_sum_pool.argtypes = [u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p]
    L.ref_round_lut_batch.argtypes = [u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(RefTier),
                                      C.POINTER(RefTier), i64p, C.c_int, C.c_void_p, u64p]
    L.ref_round_lut_batch.restype = C.c_int
    L.ref_pair_secret.argtypes = [u8p, C.c_int, u8p]
    L.ref_pbs_mb2_batch.argtypes = [u64p, C.c_int, C.c_int, u64p, C.c_int, C.c_int, C.c_int, C.c_int, i64p, C.c_int, C.c_void_p,
                                    C.c_int, u64p]
    L.ref_pbs_mb2_batch.restype = C.c_int
    L.ref_num_threads.restype = C.c_int
    L.ref_set_num_threads.argtypes = [C.c_int]
    _LIB = L
    return L


# ----------------------------------------------------------------------------- thin numpy wrappers
def gen_binary_key(seed, length):
    k = np.zeros(length, np.uint8)
    lib().ref_gen_binary_key(seed, length, k)
    return k


def lwe_encrypt(key, D, phases, sigma, seed, dim_eff=None):
    phases = np.ascontiguousarray(phases, np.uint64)
    out = np.zeros((phases.size, D + 1), np.uint64)
    lib().ref_lwe_encrypt_batch(key, D, D if dim_eff is None else dim_eff, phases, phases.size, float(sigma), seed, out)
    return out


def lwe_phase(key, D, cts):
    cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, D + 1)
    out = np.zeros(cts.shape[0], np.uint64)
    lib().ref_lwe_phase_batch(key, D, cts, cts.shape[0], out)
    return out


def ksk_gen(S_big, s_small, lk, betak, sigma, seed):
    D, n = S_big.size, s_small.size
    ksk = np.zeros((D, lk, n + 1), np.uint64)
    lib().ref_ksk_gen(S_big, D, s_small, n, lk, betak, float(sigma), seed, ksk)
    return ksk


def bsk_gen(s_small, S_glwe, k, N, l, beta, sigma, seed):
    n = s_small.size
    bsk = np.zeros((n, (k + 1) * l, k + 1, N), np.uint64)
    lib().ref_bsk_gen(s_small, n, np.ascontiguousarray(S_glwe[: k * N]), k, N, l, beta, float(sigma), seed, bsk)
    return bsk


def bsk_to_fourier(bsk):
    n, rows, kp1, N = bsk.shape
    out = np.zeros((n, rows, kp1, N), np.float64)
    lib().ref_bsk_to_fourier(bsk, n, kp1 - 1, N, rows // kp1, out)
    return out


def keyswitch(cts, ksk, betak):
    D, lk, n1 = ksk.shape
    cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, D + 1)
    out = np.zeros((cts.shape[0], n1), np.uint64)
    lib().ref_keyswitch(cts, cts.shape[0], D, ksk, n1 - 1, lk, betak, out)
    return out


def pbs(cts_small, bsk_f, bsk, k, N, l, beta, tables, w, table_idx, D_out, exact=False):
    cts_small = np.ascontiguousarray(cts_small, np.uint64)
    count, n1 = cts_small.shape
    tables = np.ascontiguousarray(tables, np.int64).reshape(-1, 1 << w)
    idx = None if table_idx is None else np.ascontiguousarray(table_idx, np.int32)
    out = np.zeros((count, D_out + 1), np.uint64)
    lib().ref_pbs_batch(cts_small, count, n1 - 1,
                        None if bsk_f is None else bsk_f.ctypes.data, None if bsk is None else bsk.ctypes.data,
                        1 if exact else 0, k, N, l, beta, tables, w,
                        None if idx is None else idx.ctypes.data, D_out, out)
    return out


def ms_center(cts_small, N):
    """centred mod switch (ref_ms_center): returns the adjusted copy"""
    out = np.ascontiguousarray(cts_small, np.uint64).copy()
    lib().ref_ms_center(out, out.shape[0], out.shape[1] - 1, N)
    return out


def pair_secret(s):
    out = np.zeros(3 * (s.size // 2), np.uint8)
    lib().ref_pair_secret(np.ascontiguousarray(s, np.uint8), s.size, out)
    return out


def pbs_mb2(cts_small, bsk3, k, N, l, beta, tables, w, table_idx, D_out):
    """two-bit blind rotation in exact arithmetic; bsk3 = standard-domain key of pair_secret(s) (3n/2 blocks)"""
    cts_small = np.ascontiguousarray(cts_small, np.uint64)
    count, n1 = cts_small.shape
    tables = np.ascontiguousarray(tables, np.int64).reshape(-1, 1 << w)
    idx = None if table_idx is None else np.ascontiguousarray(table_idx, np.int32)
    out = np.zeros((count, D_out + 1), np.uint64)
    rc = lib().ref_pbs_mb2_batch(cts_small, count, n1 - 1, np.ascontiguousarray(bsk3, np.uint64), k, N, l, beta, tables, w,
                                 None if idx is None else idx.ctypes.data, D_out, out)
    assert rc == 0
    return out


def make_tier(n, k, N, l, beta, lk, betak, bsk_f, ksk):
    t = RefTier(n, k, N, l, beta, lk, betak, bsk_f.ctypes.data, ksk.ctypes.data)
    t._keep = (bsk_f, ksk)
    return t


def round_lut(cts, D, p, r, bit_tier, tab_tier, tables, w, table_idx):
    cts = np.ascontiguousarray(cts, np.uint64).reshape(-1, D + 1)
    tables = np.ascontiguousarray(tables, np.int64).reshape(-1, 1 << w)
    idx = None if table_idx is None else np.ascontiguousarray(table_idx, np.int32)
    out = np.zeros_like(cts)
    bt = C.byref(bit_tier) if bit_tier is not None else None
    lib().ref_round_lut_batch(cts, cts.shape[0], D, p, r, bt, C.byref(tab_tier), tables, w,
                              None if idx is None else idx.ctypes.data, out)
    return out


def conv2d(cts, Cin, H, W, D, weight, stride, pad):
    weight = np.ascontiguousarray(weight, np.int32)
    Cout, _, KH, KW = weight.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = np.zeros((Cout, Ho, Wo, D + 1), np.uint64)
    lib().ref_conv2d(np.ascontiguousarray(cts, np.uint64), Cin, H, W, D, weight, Cout, KH, KW, stride, pad, out)
    return out
