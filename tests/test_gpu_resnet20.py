"""BASELINE config #1/#2 parity: ResNet-20 24x16^2 DCT trunk, full-size exact-evaluation tiers.
decrypt(run(encrypt(q))) must equal the noise-free integer circuit on every one of the 64 outputs."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


@pytest.fixture(scope="module")
def r20():
    from dctfhe import models
    from dctfhe.quantized_module import compile_brevitas_qat_model
    from dctfhe.synthetic import synthetic_dct_batch
    calib = synthetic_dct_batch(100, seed=7)
    qm = compile_brevitas_qat_model(models.ResNet20QAT(bit_width=4, in_channels=24, img_size=16), calib, n_bits=5, rounding_threshold_bits=6,
                                    p_error=0.01)
    yield qm
    qm.close()


def _oracle(qm, q):
    from oracle import circuit_ref
    out, ov = circuit_ref.run_clear(qm.compiled.blob, qm.encode_input(q))
    assert not ov
    return qm.decode_output(out)


def test_clear_mode_eight_images(r20):
    from dctfhe.synthetic import synthetic_dct_batch
    x = synthetic_dct_batch(8, seed=42)
    q = r20.quantize_input(x)
    assert np.array_equal(r20.forward_quantized(q, "disable"), _oracle(r20, q))


def test_execute_eight_images_bit_exact(r20):
    """BASELINE config #2 AS SPECIFIED: the 8-image synthetic batch encrypted in ONE session on one GPU -- all 8 x 64 outputs equal to the
    integer circuit and all eight predicted labels equal to the clear circuit's (round 2 ran two of the eight)"""
    from dctfhe import models
    from dctfhe.synthetic import centre_classifier, synthetic_dct_batch
    x = synthetic_dct_batch(8, seed=42)
    q = r20.quantize_input(x)
    want = _oracle(r20, q)
    model = models.ResNet20QAT(bit_width=4, in_channels=24, img_size=16)
    centre_classifier(model, r20.forward(synthetic_dct_batch(32, seed=7), fhe="disable"))
    labels = (r20.dequantize_output(want) @ model.classifier_w.T + model.classifier_b).argmax(axis=1)
    assert len(set(labels.tolist())) >= 3, labels
    r20.fhe_circuit.keygen(seed=1)
    got = r20.forward_quantized(q, "execute")
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "resnet20_execute_timing.json"), "w") as f:
        json.dump(dict(r20.last_timing, images=8, io=r20.last_io), f)
    assert got.shape == (8, 64)
    assert np.array_equal(got, want), np.argwhere(got != want)
    assert np.array_equal((r20.dequantize_output(got) @ model.classifier_w.T + model.classifier_b).argmax(axis=1), labels)
    # the batch travelled in the compact wire form: 24*16*16 input rows of input_dim + 1 = 2049 words per image (not D + 1 = 8193); the 64
    # outputs come off the 6-bit pooling table's N = 8192 ring, so their rows are full width
    in_dim, out_dim = r20._session("execute", 8).dims()
    assert (in_dim, out_dim) == (2048, 8192)
    assert r20.last_io["input_bytes"] == 8 * 6144 * (in_dim + 1) * 8 and r20.last_io["output_bytes"] == 8 * 64 * (out_dim + 1) * 8
