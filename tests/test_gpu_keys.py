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
        # ... and neither do two ranks that share the key but took their own counter ranges
        b.set_encrypt_counter(1 << 32)
        assert not np.any(b.encrypt(ph)[:, :8] == c1[:, :8])
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
        # a corrupted / truncated blob is refused, not half-loaded
        from dctfhe._lib import DctfheError
        with pytest.raises(DctfheError, match="blob"):
            server.load_evaluation_keys(blob[:-8])
        bad = blob.copy()
        bad[0] ^= 0xFF
        with pytest.raises(DctfheError, match="magic"):
            server.load_evaluation_keys(bad)
    finally:
        server.close()
        client.close()
