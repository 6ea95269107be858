"""The noise model the compiler budgets with (dctfhe/params.py) against what the GPU bootstrap really leaves on its
output, for every full-size tier of the default catalogue.  This is the calibration of the f64-FFT error term
(var ~ c (k+1) l N^2 B^2 / 12 * 2^-106 per CMUX, c = 2): measured sigma must sit within [0.5x, 1.6x] of the model."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cent(x):
    return x.astype(np.int64).astype(np.float64) / 2.0 ** 64


def test_output_noise_matches_model(gpu_ctx):
    from dctfhe import params as P
    from dctfhe.engine import Keys
    ps = P.default_params()
    keys = Keys(gpu_ctx, P.to_c_params(ps), seed=3)
    try:
        rng = np.random.default_rng(0)
        count = 1024
        report = {}
        for ti, t in enumerate(ps.tiers):
            one_bit = t.name.startswith("B")
            if one_bit:       # margin 1/4 tiers: feed sign inputs
                bits = rng.integers(0, 2, count).astype(np.uint64)
                cts = keys.encrypt(bits << np.uint64(63))
                table, w = np.array([1 << 57], np.int64), 0
                want = np.where(bits == 0, np.int64(1 << 57), np.int64(-(1 << 57))).astype(np.int64).view(np.uint64)
            else:
                msgs = rng.integers(0, 8, count).astype(np.uint64)
                cts = keys.encrypt(msgs << np.uint64(60))
                table, w = np.arange(8, dtype=np.int64) << 57, 3
                want = msgs << np.uint64(57)
            out = keys.pbs(ti, keys.keyswitch(ti, cts), table, w)
            err = _cent(keys.decrypt(out) - want)
            assert np.abs(err).max() < 2.0 ** -9, (t.name, "wrong outputs")
            measured, model = err.std(), math.sqrt(P.var_pbs_out(t, ps.fft_noise_c))
            report[t.name] = (math.log2(measured), math.log2(model))
            assert 0.5 * model < measured < 1.6 * model, (t.name, math.log2(measured), math.log2(model))
        print("sigma_out log2 (measured, model):", {k: (round(a, 2), round(b, 2)) for k, (a, b) in report.items()})
    finally:
        keys.close()


def test_keyswitch_noise_matches_model(gpu_ctx, oracle):
    """the other half of every site's budget: what the key switch of each tier (its own gadget base, depth and small-key length) adds to
    a fresh ciphertext, against params.var_keyswitch over the effective dimension (the client masks the first input_dim key bits only)"""
    from dctfhe import params as P
    from dctfhe.engine import Keys
    ps = P.default_params()
    keys = Keys(gpu_ctx, P.to_c_params(ps), seed=5)
    try:
        _, s = keys.export_secret()
        rng = np.random.default_rng(1)
        msgs = rng.integers(0, 16, 4096).astype(np.uint64) << np.uint64(59)
        cts = keys.encrypt(msgs)
        deff = ps.input_dim or ps.D
        report = {}
        for ti, t in enumerate(ps.tiers):
            small = keys.keyswitch(ti, cts)
            err = _cent(oracle.lwe_phase(s[:t.n].copy(), t.n, small) - msgs)
            measured, model = err.std(), math.sqrt(P.var_keyswitch(deff, t) + ps.input_sigma ** 2)
            report[t.name] = (round(math.log2(measured), 2), round(math.log2(model), 2))
            assert 0.7 * model < measured < 1.3 * model, (t.name, report[t.name])
        print("sigma after key switch log2 (measured, model):", report)
    finally:
        keys.close()
