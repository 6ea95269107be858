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


def test_centred_mod_switch_halves_the_rounding_noise(gpu_ctx, oracle):
    """k_ms_center (the scheduler runs it between every key switch and its bootstrap): bit-exact against the oracle's ref_ms_center, and the
    error of the switched phase -- b~ - sum a~_i s_i against the exact phase, in units of the 2N levels -- has the variance the compiler
    budgets with: n/4 + 1 twelfths per level^2 centred against n/2 + 1 plain (params.var_modswitch), measured with the real small key
    on key-switched ciphertexts of the 6-bit tier's shape (N = 8192, n = 800) and of the one-bit tiers' (N = 1024, n = 560)."""
    from dctfhe import params as P
    from dctfhe.engine import Keys
    ps = P.default_params()
    keys = Keys(gpu_ctx, P.to_c_params(ps), seed=9)
    try:
        _, s = keys.export_secret()
        rng = np.random.default_rng(2)
        msgs = rng.integers(0, 16, 4096).astype(np.uint64) << np.uint64(59)
        cts = keys.encrypt(msgs)
        names = [t.name for t in ps.tiers]
        for name in ("T6a", "Ba2"):
            ti = names.index(name)
            t = ps.tiers[ti]
            small = keys.keyswitch(ti, cts)
            centred = keys.modswitch_center(ti, small)
            assert np.array_equal(centred, oracle.ms_center(small, t.N))                       # integer arithmetic: bit for bit
            assert np.array_equal(centred[:, :t.n], small[:, :t.n]) and np.any(centred[:, t.n] != small[:, t.n])
            sk = s[:t.n].astype(np.int64)
            sh = np.uint64(63 - t.logN)

            def switched_phase_error(c):
                lv = ((c >> (sh - np.uint64(1))) + np.uint64(1)) >> np.uint64(1)               # round(word / 2^sh): the 2N levels (before the mask)
                exact = (c[:, t.n] - (c[:, :t.n] * sk.astype(np.uint64)).sum(axis=1, dtype=np.uint64)).astype(np.uint64)
                coarse = ((lv[:, t.n] - (lv[:, :t.n] * sk.astype(np.uint64)).sum(axis=1, dtype=np.uint64)) << sh).astype(np.uint64)
                return _cent(coarse - exact) * 2.0 * t.N                                       # in levels

            e_plain = switched_phase_error(small)
            # the centred body differs from the plain one by R/2: compare its switched phase with the ORIGINAL exact phase
            lv = ((centred >> (sh - np.uint64(1))) + np.uint64(1)) >> np.uint64(1)
            exact = (small[:, t.n] - (small[:, :t.n] * sk.astype(np.uint64)).sum(axis=1, dtype=np.uint64)).astype(np.uint64)
            coarse = ((lv[:, t.n] - (lv[:, :t.n] * sk.astype(np.uint64)).sum(axis=1, dtype=np.uint64)) << sh).astype(np.uint64)
            e_cent = _cent(coarse - exact) * 2.0 * t.N
            h = int(sk.sum())
            v_plain, v_cent = (h + 1) / 12.0, (t.n / 4.0 + 1) / 12.0                            # per level^2 (the model assumes h = n/2)
            assert 0.85 * v_plain < e_plain.var() < 1.15 * v_plain, (name, e_plain.var(), v_plain)
            assert 0.85 * v_cent < e_cent.var() < 1.15 * v_cent, (name, e_cent.var(), v_cent)
            assert abs(e_cent.mean()) < 4 * math.sqrt(v_cent / e_cent.size) + 0.6                # |h - n/2| leaves a small mean: (h - n/2) E[e] = 0
            model = P.var_modswitch(t) * (2.0 * t.N) ** 2
            assert abs(model - v_cent) < 1e-9
    finally:
        keys.close()
