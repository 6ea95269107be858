"""Client / server key separation (reference call site homomorphic_eval.py:313-317):
the client generates keys and ships the evaluation keys as a blob; a server-side module imports the blob, evaluates
ciphertexts with it and holds nothing secret; the client decrypts.  Plus the CSPRNG on the device and key determinism."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_out(qm, q):
    from oracle import circuit_ref
    out, ov = circuit_ref.run_clear(qm.compiled.blob, qm.encode_input(q))
    assert not ov
    return qm.decode_output(out)


def test_device_generator_matches_host(gpu_ctx):
    L = gpu_ctx.L
    key = bytes(range(7, 39))
    a, b = np.zeros(1000, np.uint64), np.zeros(1000, np.uint64)
    assert L.dctfhe_rng_host(key, 300, 5, 1000, a.ctypes.data_as(C.c_void_p)) == 0
    assert L.dctfhe_rng_device(gpu_ctx.h, key, 300, 5, 1000, b.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(a, b)


def test_keys_are_a_function_of_the_seed(gpu_ctx):
    from dctfhe import params as P
    from dctfhe.engine import ClientKey
    cp = P.to_c_params(P.test_params())
    a, b, c = ClientKey(gpu_ctx, cp, 5), ClientKey(gpu_ctx, cp, 5), ClientKey(gpu_ctx, cp, 6)
    try:
        Sa, sa = a.export_secret()
        Sb, sb = b.export_secret()
        Sc, sc = c.export_secret()
        assert np.array_equal(Sa, Sb) and np.array_equal(sa, sb) and not np.array_equal(Sa, Sc)
        assert 0.4 < Sa.mean() < 0.6 and set(np.unique(Sa)) == {0, 1}
        ea, eb = a.generate_eval_keys(), b.generate_eval_keys()
        assert np.array_equal(ea.export_ksk(0), eb.export_ksk(0))                # what every rank of a job relies on
        assert np.array_equal(a.export_bsk(1), b.export_bsk(1))
        # two encryptions of the same phases under one key never share masks or noise
        ph = np.arange(4, dtype=np.uint64) << np.uint64(58)
        c1, c2 = a.encrypt(ph), a.encrypt(ph)
        assert not np.any(c1[:, :8] == c2[:, :8]) and np.array_equal((a.decrypt(c1) + np.uint64(1 << 56)) >> np.uint64(57), (a.decrypt(c2) + np.uint64(1 << 56)) >> np.uint64(57))
        # ... and neither do two HANDLES made from one seed (a re-created key, another process, the ranks of a job), by default and
        # at the same call counter: the encryption streams hang on a per-handle nonce from the OS, not on the seed alone (ADVICE r2:
        # before, b.encrypt re-drew a's masks and noise, and ct_a - ct_b = (0, m_a - m_b) gave away plaintext differences)
        b.set_encrypt_counter(0)
        a2 = ClientKey(gpu_ctx, cp, 5)
        cb, ca2 = b.encrypt(ph), a2.encrypt(ph)
        assert not np.any(cb[:, :8] == c1[:, :8]) and not np.any(ca2[:, :8] == c1[:, :8]) and not np.any(ca2[:, :8] == cb[:, :8])
        assert not np.any(cb[:, -1] == c1[:, -1]) and not np.any(ca2[:, -1] == c1[:, -1])            # bodies: other masks AND other noise
        rnd = lambda x: (x + np.uint64(1 << 56)) >> np.uint64(57)
        assert np.array_equal(rnd(a.decrypt(cb)), rnd(a.decrypt(c1))) and np.array_equal(rnd(a.decrypt(ca2)), rnd(ph))       # same key all the same
        # only a caller who FIXES the nonce (reproducible experiments) gets equal ciphertexts from equal (seed, nonce, counter)
        for h in (a2, b):
            h.set_encrypt_nonce(bytes(range(16)))
            h.set_encrypt_counter(7)
        assert np.array_equal(a2.encrypt(ph), b.encrypt(ph))
        a2.set_encrypt_nonce(bytes(range(1, 17)))
        a2.set_encrypt_counter(8)
        assert not np.any(a2.encrypt(ph)[:, :8] == b.encrypt(ph)[:, :8])
        a2.close()
        # the compact wire form holds the same ciphertexts as the full-width form of the same call
        D, dim = a.D, a.input_dim
        for h in (a, b):
            h.set_encrypt_nonce(bytes(range(16)))
            h.set_encrypt_counter(20)
        full, compact = a.encrypt(ph), b.encrypt(ph, dim)
        assert compact.shape == (4, dim + 1) and np.array_equal(full[:, :dim], compact[:, :dim]) and np.array_equal(full[:, D], compact[:, dim])
        assert not full[:, dim:D].any() and np.array_equal(a.decrypt(full), a.decrypt(compact, dim))
        ea.close(); eb.close()
    finally:
        a.close(); b.close(); c.close()


def test_server_evaluates_with_imported_evaluation_keys():
    from dctfhe import models, params as P
    from dctfhe.quantized_module import QuantizedModule, compile_brevitas_qat_model
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    client = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits=6, param_set=P.test_params())
    server = QuantizedModule(client.compiled)            # same compiled circuit, its own context, no keys of its own
    try:
        client.fhe_circuit.keygen()                      # 32 bytes from the OS
        blob = client.export_evaluation_keys()
        assert blob.dtype == np.uint8 and blob.size > 1 << 20
        server.load_evaluation_keys(blob)
        assert not hasattr(server._keys, "client")       # the server handle has no secret half
        q = client.quantize_input(calib[:3])
        cts_in = client._keys.client.encrypt(client.encode_input(q).reshape(-1))
        cts_out = server.evaluate_encrypted(cts_in, batch=3)
        got = client.decode_output(client._keys.client.decrypt(cts_out).reshape(3, -1))
        assert np.array_equal(got, _oracle_out(client, q))
        # the same exchange in the compact wire form: input rows of input_dim + 1 words, output rows of the last tier's ring + 1
        dim = client._keys.client.input_dim
        cts_c = client._keys.client.encrypt(client.encode_input(q).reshape(-1), dim)
        out_c = server.evaluate_encrypted(cts_c, batch=3, dim=dim)
        out_dim = out_c.shape[1] - 1
        assert out_dim == server._session("execute", 3).dims()[1] <= client._keys.D and cts_c.shape[1] == dim + 1
        assert np.array_equal(client.decode_output(client._keys.client.decrypt(out_c, out_dim).reshape(3, -1)), _oracle_out(client, q))
        # a corrupted / truncated blob is refused, not half-loaded
        from dctfhe._lib import DctfheError
        with pytest.raises(DctfheError, match="blob"):
            server.load_evaluation_keys(blob[:-8])
        bad = blob.copy()
        bad[0] ^= 0xFF
        with pytest.raises(DctfheError, match="magic"):
            server.load_evaluation_keys(bad)
        # a key-switch key off its torus grid (the matrix-core form keeps the top ks_limbs bytes of a word only) is refused too
        from dctfhe import _lib
        off = blob.copy()
        off[16 + C.sizeof(_lib.Params)] ^= 1                      # lowest bit of the first key-switch-key word of tier 0
        with pytest.raises(DctfheError, match="torus grid"):
            server.load_evaluation_keys(off)
        server.load_evaluation_keys(blob)                         # (and the good blob still loads afterwards)
    finally:
        server.close()
        client.close()


def test_key_switch_keys_live_on_their_torus_grid(gpu_ctx, oracle):
    """the key-switch key of a tier keeps ks_limbs = 2 (one-bit tiers: 5 levels of base 4) or 4 (table tiers: 9 levels) top bytes per
    word -- masks drawn on that grid, body rounded to it -- so that the i8 matrix-core GEMM multiplies 2 / 4 byte limbs per word instead
    of 8.  The exported key IS that key (low bytes zero), each row still an encryption of S_i / B^(lev+1) under the small key with
    the tier's noise, and the device key switch equals the oracle's on it bit for bit."""
    from dctfhe import params as P
    from dctfhe.engine import Keys
    ps = P.default_params()
    keys = Keys(gpu_ctx, P.to_c_params(ps), seed=11)
    try:
        S, s = keys.export_secret()
        rng = np.random.default_rng(4)
        cts = keys.encrypt(rng.integers(0, 16, 64).astype(np.uint64) << np.uint64(59))
        seen = set()
        for ti, t in enumerate(ps.tiers):
            limbs = P.ks_limbs(t)
            seen.add(limbs)
            ksk = keys.export_ksk(ti)
            low = np.uint64((1 << (64 - 8 * limbs)) - 1)
            assert not (ksk & low).any() and (ksk & (low + np.uint64(1))).any(), t.name        # on the grid, and the grid's last bit is used
            rows = ksk[:256].reshape(-1, t.n + 1)                                               # the first 256 key bits, every level
            ph = oracle.lwe_phase(s[:t.n].copy(), t.n, rows).reshape(256, t.lk)
            for lev in range(t.lk):
                want = S[:256].astype(np.uint64) << np.uint64(64 - t.betak * (lev + 1))
                err = (ph[:, lev] - want).astype(np.int64).astype(np.float64) / 2.0 ** 64
                assert np.abs(err).max() < 6 * t.lwe_sigma + 2.0 ** (-8 * limbs), (t.name, lev)
            small = keys.keyswitch(ti, cts)
            assert np.array_equal(small, oracle.keyswitch(cts, ksk, t.betak)), t.name
        assert seen == {2, 4}
    finally:
        keys.close()
