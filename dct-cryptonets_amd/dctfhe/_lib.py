"""ctypes binding of libdctfhe.so (include/dctfhe.h).

The library is built in-tree by __graft_entry__.build() (hipcc --offload-arch=gfx950) and lives at
dct-cryptonets_amd/libdctfhe.so.  There is no fallback: if the library is missing, or no HIP
device is visible, every operation raises.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "libdctfhe.so")
MAX_TIERS = 12


class Tier(C.Structure):
    _fields_ = [("n", C.c_int32), ("k", C.c_int32), ("logN", C.c_int32), ("l", C.c_int32), ("beta", C.c_int32),
                ("lk", C.c_int32), ("betak", C.c_int32), ("ksk_share", C.c_int32),
                ("unroll", C.c_int32), ("key_lds", C.c_int32),
                ("lwe_sigma", C.c_double), ("glwe_sigma", C.c_double)]


class Params(C.Structure):
    _fields_ = [("D", C.c_int32), ("n_max", C.c_int32), ("n_tiers", C.c_int32), ("input_dim", C.c_int32),
                ("input_sigma", C.c_double), ("tiers", Tier * MAX_TIERS)]


class Stats(C.Structure):
    _fields_ = [("pbs_count", C.c_int64 * MAX_TIERS), ("ks_count", C.c_int64 * MAX_TIERS), ("conv_macs", C.c_int64),
                ("lut_sites", C.c_int64), ("bit_steps", C.c_int64), ("bytes_algorithmic", C.c_double),
                ("key_bytes_per_pass", C.c_double), ("flops_f64", C.c_double), ("max_bit_width", C.c_int32),
                ("n_ops", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("pbs_ms", C.c_double * MAX_TIERS), ("ks_ms", C.c_double),
                ("linear_ms", C.c_double), ("pbs_launches", C.c_int64 * MAX_TIERS), ("pbs_cts", C.c_int64 * MAX_TIERS)]


EXPORTS = [
    "dctfhe_last_error", "dctfhe_version", "dctfhe_ctx_create", "dctfhe_ctx_destroy", "dctfhe_ctx_set_stream",
    "dctfhe_ctx_synchronize", "dctfhe_keygen", "dctfhe_client_key_create", "dctfhe_client_key_destroy", "dctfhe_eval_keys_generate",
    "dctfhe_eval_keys_destroy", "dctfhe_eval_keys_export", "dctfhe_eval_keys_import", "dctfhe_client_key_export_secret",
    "dctfhe_eval_keys_export_ksk", "dctfhe_client_key_export_bsk", "dctfhe_rng_host", "dctfhe_rng_device", "dctfhe_client_key_set_encrypt_counter", "dctfhe_client_key_set_encrypt_nonce", "dctfhe_encrypt", "dctfhe_decrypt", "dctfhe_encrypt_rows", "dctfhe_decrypt_rows", "dctfhe_keyswitch", "dctfhe_keyswitch_prefix", "dctfhe_session_set_noise",
    "dctfhe_pbs", "dctfhe_modswitch_center", "dctfhe_round_lut", "dctfhe_conv2d", "dctfhe_add_rows", "dctfhe_affine_rows", "dctfhe_sum_pool_rows", "dctfhe_circuit_load", "dctfhe_circuit_destroy", "dctfhe_params_check", "dctfhe_circuit_validate", "dctfhe_dct_frontend",
    "dctfhe_circuit_stats", "dctfhe_circuit_io", "dctfhe_session_create", "dctfhe_session_destroy",
    "dctfhe_session_upload", "dctfhe_session_run", "dctfhe_session_download", "dctfhe_session_upload_rows", "dctfhe_session_download_rows", "dctfhe_session_dims", "dctfhe_fp64_peak", "dctfhe_bench_pbs",
]

_lib = None


class DctfheError(RuntimeError):
    pass


def load():
    """Load libdctfhe.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DctfheError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); this engine has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, sz = C.c_void_p, C.c_uint64, C.c_int, C.c_size_t
    L.dctfhe_last_error.restype = C.c_char_p
    L.dctfhe_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.dctfhe_ctx_destroy.argtypes = [vp]
    L.dctfhe_ctx_set_stream.argtypes = [vp, vp]
    L.dctfhe_ctx_synchronize.argtypes = [vp]
    L.dctfhe_keygen.argtypes = [vp, C.POINTER(Params), C.c_char_p, C.POINTER(vp), C.POINTER(vp)]
    L.dctfhe_client_key_create.argtypes = [vp, C.POINTER(Params), C.c_char_p, C.POINTER(vp)]
    L.dctfhe_client_key_destroy.argtypes = [vp]
    L.dctfhe_eval_keys_generate.argtypes = [vp, C.POINTER(vp)]
    L.dctfhe_eval_keys_destroy.argtypes = [vp]
    L.dctfhe_eval_keys_export.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.dctfhe_eval_keys_import.argtypes = [vp, vp, sz, C.POINTER(vp)]
    L.dctfhe_client_key_export_secret.argtypes = [vp, vp, vp]
    L.dctfhe_eval_keys_export_ksk.argtypes = [vp, i32, vp]
    L.dctfhe_client_key_export_bsk.argtypes = [vp, i32, vp]
    L.dctfhe_rng_host.argtypes = [C.c_char_p, u64, u64, sz, vp]
    L.dctfhe_rng_device.argtypes = [vp, C.c_char_p, u64, u64, sz, vp]
    L.dctfhe_client_key_set_encrypt_counter.argtypes = [vp, u64]
    L.dctfhe_client_key_set_encrypt_nonce.argtypes = [vp, C.c_char_p]
    L.dctfhe_encrypt.argtypes = [vp, vp, vp, sz, vp]
    L.dctfhe_decrypt.argtypes = [vp, vp, vp, sz, vp]
    L.dctfhe_encrypt_rows.argtypes = [vp, vp, vp, sz, i32, vp]
    L.dctfhe_decrypt_rows.argtypes = [vp, vp, vp, sz, i32, vp]
    L.dctfhe_keyswitch.argtypes = [vp, vp, i32, vp, sz, i32, vp]
    L.dctfhe_session_set_noise.argtypes = [vp, C.c_uint64, vp, i32]
    L.dctfhe_keyswitch_prefix.argtypes = [vp, vp, i32, vp, sz, i32, i32, vp]
    L.dctfhe_modswitch_center.argtypes = [vp, vp, i32, vp, sz]
    L.dctfhe_pbs.argtypes = [vp, vp, i32, vp, sz, vp, i32, i32, vp, vp]
    L.dctfhe_round_lut.argtypes = [vp, vp, i32, i32, vp, sz, i32, i32, vp, i32, i32, vp, vp]
    L.dctfhe_conv2d.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp]
    L.dctfhe_add_rows.argtypes = [vp, vp, i32, i32, vp, i32, i32, sz, i32, vp]
    L.dctfhe_affine_rows.argtypes = [vp, vp, i32, i32, sz, i32, i32, u64, i32, vp]
    L.dctfhe_sum_pool_rows.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.dctfhe_dct_frontend.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp]
    L.dctfhe_params_check.argtypes = [C.POINTER(Params)]
    L.dctfhe_circuit_validate.argtypes = [vp, sz]
    L.dctfhe_circuit_load.argtypes = [vp, vp, sz, C.POINTER(vp)]
    L.dctfhe_circuit_destroy.argtypes = [vp]
    L.dctfhe_circuit_stats.argtypes = [vp, C.POINTER(Params), C.POINTER(Stats)]
    L.dctfhe_circuit_io.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dctfhe_session_create.argtypes = [vp, vp, vp, i32, C.POINTER(vp)]
    L.dctfhe_session_destroy.argtypes = [vp]
    L.dctfhe_session_upload.argtypes = [vp, vp]
    L.dctfhe_session_run.argtypes = [vp, C.POINTER(Timing)]
    L.dctfhe_session_download.argtypes = [vp, vp]
    L.dctfhe_session_upload_rows.argtypes = [vp, vp, i32]
    L.dctfhe_session_download_rows.argtypes = [vp, vp, i32]
    L.dctfhe_session_dims.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dctfhe_fp64_peak.argtypes = [vp, C.POINTER(C.c_double)]
    L.dctfhe_bench_pbs.argtypes = [vp, vp, i32, sz, i32, C.POINTER(C.c_double)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise DctfheError(load().dctfhe_last_error().decode())


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
