"""Other BASELINE configs as parity cases: the circuit compiled for them, evaluated noise-free on the GPU scheduler
(1-word ciphertexts) against the numpy integer circuit.  Encrypted runs of these sizes take minutes per image and
are exercised by bench.py --config instead."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rgb_batch(n, seed, size=32):
    from dctfhe import frontend, synthetic
    tf = frontend.rgb_eval_transform(size)
    return np.stack([tf(im) for im in synthetic.synthetic_images(n, seed)]).astype(np.float32)


@pytest.mark.parametrize("name,fn,conv_outputs,feat", [("R20 3x32^2 (config #3)", "ResNet20QAT", 860160, 256), ("R18 3x32^2 (config #4)", "ResNet18QAT", 614400, 512)])
def test_clear_circuit_matches_oracle(name, fn, conv_outputs, feat):
    from dctfhe import compile as cc, models
    from dctfhe.quantized_module import QuantizedModule
    from oracle import circuit_ref
    model = getattr(models, fn)(bit_width=4, in_channels=3, img_size=32)
    compiled = cc.compile_model(model, _rgb_batch(64, 7))      # the reference calibrates on 64-100 images (io_utils.py:72)
    convs = sum(compiled.tensors[o.dst].C * compiled.tensors[o.dst].H * compiled.tensors[o.dst].W for o in compiled.ops if o.type == cc.OP_CONV)
    assert convs == conv_outputs                      # SURVEY 8a table "Other BASELINE configs"
    assert compiled.n_out() == feat == model.final_feat_dim
    qm = QuantizedModule(compiled)
    try:
        q = qm.quantize_input(_rgb_batch(2, 42))
        ref, overflow = circuit_ref.run_clear(compiled.blob, qm.encode_input(q))
        assert not overflow
        assert np.array_equal(qm.forward_quantized(q, "disable"), qm.decode_output(ref))
    finally:
        qm.close()
