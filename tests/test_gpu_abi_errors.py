"""One negative case per C-ABI entry point on the device: bad arguments come back as an error code and
a message, nothing crashes, and the handle is usable afterwards (the early returns release what they had allocated)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kit(gpu_ctx):
    from dctfhe import compile as cc, models, params as P
    from dctfhe.engine import Circuit, Keys
    ps = P.test_params()
    keys = Keys(gpu_ctx, P.to_c_params(ps), seed=2)
    calib = np.random.default_rng(0).normal(0, 1, (16, 4, 6, 6))
    compiled = cc.compile_model(models.tiny_resnet_q(), calib, param_set=ps)
    circ = Circuit(gpu_ctx, compiled.blob)
    yield gpu_ctx, keys, circ, compiled, ps
    circ.close()
    keys.close()


def _fails(L, rc, needle):
    assert rc != 0
    msg = L.dctfhe_last_error().decode()
    assert needle in msg, msg


def test_context_and_keygen_errors(kit):
    ctx, keys, circ, compiled, ps = kit
    L = ctx.L
    h = C.c_void_p()
    _fails(L, L.dctfhe_ctx_create(99, C.byref(h)), "out of range")
    from dctfhe import params as P
    bad = P.to_c_params(ps)
    bad.tiers[0].l = 7
    c, e = C.c_void_p(), C.c_void_p()
    _fails(L, L.dctfhe_keygen(ctx.h, C.byref(bad), bytes(32), C.byref(c), C.byref(e)), "bad bootstrap gadget")
    _fails(L, L.dctfhe_client_key_create(ctx.h, C.byref(bad), bytes(32), C.byref(c)), "bad bootstrap gadget")
    _fails(L, L.dctfhe_client_key_create(ctx.h, C.byref(P.to_c_params(ps)), None, C.byref(c)), "null")
    _fails(L, L.dctfhe_eval_keys_generate(None, C.byref(e)), "null")
    n = C.c_size_t()
    small = np.zeros(64, np.uint8)
    _fails(L, L.dctfhe_eval_keys_export(keys.eval.h, small.ctypes.data_as(C.c_void_p), small.size, C.byref(n)), "needed")
    _fails(L, L.dctfhe_eval_keys_import(ctx.h, small.ctypes.data_as(C.c_void_p), small.size, C.byref(e)), "too short")
    _fails(L, L.dctfhe_client_key_set_encrypt_counter(keys.client.h, 1 << 60), "out of range")


def test_primitive_errors(kit):
    ctx, keys, circ, compiled, ps = kit
    L = ctx.L
    D = keys.D
    cts = keys.encrypt(np.arange(3, dtype=np.uint64) << np.uint64(58))
    out_small = np.zeros((3, 64), np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _fails(L, L.dctfhe_keyswitch(ctx.h, keys.eval.h, 5, p(cts), 3, 0, p(out_small)), "tier out of range")
    _fails(L, L.dctfhe_keyswitch_prefix(ctx.h, keys.eval.h, 0, p(cts), 3, 0, D + 4, p(out_small)), "deff out of range")
    _fails(L, L.dctfhe_keyswitch(ctx.h, keys.eval.h, 0, p(cts), 3, 64, p(out_small)), "shift out of range")
    tab = np.zeros(1 << 12, np.int64)
    small = np.zeros((3, ps.tiers[0].n + 1), np.uint64)
    big = np.zeros((3, D + 1), np.uint64)
    _fails(L, L.dctfhe_pbs(ctx.h, keys.eval.h, 0, p(small), 3, p(tab), 1, 12, None, p(big)), "does not fit")
    _fails(L, L.dctfhe_pbs(ctx.h, keys.eval.h, -1, p(small), 3, p(tab), 1, 4, None, p(big)), "tier out of range")
    idx = np.array([0, 3, 0], np.int32)
    _fails(L, L.dctfhe_pbs(ctx.h, keys.eval.h, 0, p(small), 3, p(tab), 2, 4, p(idx), p(big)), "table_idx[1]")
    _fails(L, L.dctfhe_round_lut(ctx.h, keys.eval.h, 1, 0, p(cts), 3, 8, 2, p(tab), 1, 5, None, p(big)), "w must equal p - r")
    _fails(L, L.dctfhe_round_lut(ctx.h, keys.eval.h, 9, 0, p(cts), 3, 8, 2, p(tab), 1, 6, None, p(big)), "tier out of range")
    _fails(L, L.dctfhe_round_lut(ctx.h, keys.eval.h, 1, 0, p(cts), 3, 70, 2, p(tab), 1, 68, None, p(big)), "p <= 62")
    w = np.zeros((2, 2, 3, 3), np.int8)
    x = np.zeros((1, 2, 2, 2, 5), np.uint64)
    _fails(L, L.dctfhe_conv2d(ctx.h, 4, p(x), 1, 2, 2, 2, p(w), 2, 3, 3, 1, 0, p(x)), "kernel larger")
    _fails(L, L.dctfhe_conv2d(ctx.h, 4, p(x), 1, 2, 2, 2, p(w), 2, 3, 3, 0, 1, p(x)), "bad geometry")
    # the handles still work after all that
    assert np.array_equal((keys.decrypt(cts) + np.uint64(1 << 56)) >> np.uint64(57), (np.arange(3, dtype=np.uint64) << np.uint64(1)))


def test_circuit_and_session_errors(kit):
    ctx, keys, circ, compiled, ps = kit
    L = ctx.L
    h = C.c_void_p()
    blob = bytearray(compiled.blob)
    _fails(L, L.dctfhe_circuit_load(ctx.h, bytes(blob[:40]), 40, C.byref(h)), "truncated")
    blob[0] ^= 1
    _fails(L, L.dctfhe_circuit_load(ctx.h, bytes(blob), len(blob), C.byref(h)), "magic")
    _fails(L, L.dctfhe_session_create(ctx.h, circ.h, keys.eval.h, 0, C.byref(h)), "batch must be")
    # a circuit that names a tier these keys lack
    from dctfhe import params as P
    from dctfhe.engine import Keys, Session
    one_tier = P.to_c_params(ps)
    one_tier.n_tiers = 1
    one_tier.n_max = one_tier.tiers[0].n
    k1 = Keys(ctx, one_tier, seed=3)
    try:
        _fails(L, L.dctfhe_session_create(ctx.h, circ.h, k1.eval.h, 1, C.byref(h)), "names a tier the keys lack")
    finally:
        k1.close()
    # the effective-dimension guard at upload: ciphertexts with a non-zero tail when the circuit was compiled for a key prefix
    sess = Session(ctx, circ, keys, 1)
    try:
        assert L.dctfhe_session_set_noise(sess.h, 1, None, 0) != 0 and "clear-mode" in L.dctfhe_last_error().decode()
    finally:
        sess.close()
    clear = Session(ctx, circ, None, 1)
    try:
        sg = np.zeros(3, np.float64)
        _fails(L, L.dctfhe_session_set_noise(clear.h, 1, sg.ctypes.data_as(C.c_void_p), 3), "one sigma per op")
    finally:
        clear.close()
