"""Every BASELINE config as a parity case on the encrypted HIP path.
  * clear-mode circuit (1-word ciphertexts on the GPU scheduler) against the numpy integer circuit, configs #3 and #4;
  * ENCRYPTED, full-size exact tiers, decrypt(run(encrypt(q))) == integer circuit on every output:
      #2  ResNet-20 24x16^2, the 8-image batch in one session  (tests/test_gpu_resnet20.py)
      #3  ResNet-20 3x32^2, one image (256 outputs)
      #4  ResNet-18 3x32^2, one image (512 outputs)
      #5  ResNet-18 48x112^2: one full-size image is ~8 minutes, so (i) the WHOLE trunk -- 1x1 stem without ReLU (reference
          models/backbone.py:555-563 `relu1: False`) and all four stages -- on a 56x56 crop of the 112x112 DCT planes (a quarter of
          the image), and (ii) the stem + first stage on a 16x16 crop under catalogue variants; same tiers, same per-site precisions.
The label check uses the seeded classifier centred on the calibration features (dctfhe.synthetic.centre_classifier):
labels differ between images, so comparing them checks something."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rgb_batch(n, seed, size=32):
    from dctfhe import frontend, synthetic
    tf = frontend.rgb_eval_transform(size)
    return np.stack([tf(im) for im in synthetic.synthetic_images(n, seed)]).astype(np.float32)


def _oracle(qm, q):
    from oracle import circuit_ref
    ref, overflow = circuit_ref.run_clear(qm.compiled.blob, qm.encode_input(q))
    assert not overflow
    return qm.decode_output(ref)


@pytest.mark.parametrize("name,fn,conv_outputs,feat", [("R20 3x32^2 (config #3)", "ResNet20QAT", 860160, 256), ("R18 3x32^2 (config #4)", "ResNet18QAT", 614400, 512)])
def test_clear_circuit_matches_oracle(name, fn, conv_outputs, feat):
    from dctfhe import compile as cc, models
    from dctfhe.quantized_module import QuantizedModule
    model = getattr(models, fn)(bit_width=4, in_channels=3, img_size=32)
    compiled = cc.compile_model(model, _rgb_batch(64, 7))      # the reference calibrates on 64-100 images (io_utils.py:72)
    convs = sum(compiled.tensors[o.dst].C * compiled.tensors[o.dst].H * compiled.tensors[o.dst].W for o in compiled.ops if o.type == cc.OP_CONV)
    assert convs == conv_outputs                      # SURVEY 8a table "Other BASELINE configs"
    assert compiled.n_out() == feat == model.final_feat_dim
    qm = QuantizedModule(compiled)
    try:
        q = qm.quantize_input(_rgb_batch(2, 42))
        assert np.array_equal(qm.forward_quantized(q, "disable"), _oracle(qm, q))
    finally:
        qm.close()


@pytest.mark.parametrize("name,fn,feat", [("R20 3x32^2 (config #3)", "ResNet20QAT", 256), ("R18 3x32^2 (config #4)", "ResNet18QAT", 512)])
def test_encrypted_one_image_bit_exact(name, fn, feat):
    from dctfhe import models
    from dctfhe.quantized_module import compile_brevitas_qat_model
    from dctfhe.synthetic import centre_classifier
    model = getattr(models, fn)(bit_width=4, in_channels=3, img_size=32)
    calib = _rgb_batch(100, 7)
    qm = compile_brevitas_qat_model(model, calib, n_bits=5, rounding_threshold_bits=6, p_error=0.01)
    try:
        centre_classifier(model, qm.forward(calib[:32], fhe="disable"))
        x = _rgb_batch(8, 42)
        q = qm.quantize_input(x)
        want = _oracle(qm, q)
        labels = (qm.dequantize_output(want) @ model.classifier_w.T + model.classifier_b).argmax(axis=1)
        assert len(set(labels.tolist())) >= 3, labels            # the label check is not degenerate
        qm.fhe_circuit.keygen(seed=1)
        got = qm.forward_quantized(q[:1], "execute")
        assert got.shape == (1, feat)
        assert np.array_equal(got, want[:1]), (got, want[:1])
        assert (qm.dequantize_output(got) @ model.classifier_w.T + model.classifier_b).argmax(axis=1)[0] == labels[0]
    finally:
        qm.close()


def test_encrypted_config5_prefix_bit_exact():
    """#5 ResNet-18 48x112^2: stem (1x1 48->64, no ReLU) + stage 1 (two 64-channel identity blocks) on a 16x16 crop."""
    from dctfhe import frontend, models, synthetic
    from dctfhe.quantized_module import compile_brevitas_qat_model
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=112, channels=48)
    planes = np.stack([tf(im) for im in synthetic.synthetic_images(18, 7, size=96)]).astype(np.float32)     # [18, 48, 112, 112]
    crops = planes[:, :, 40:56, 40:56]
    model = models.ResNet18QAT(bit_width=4, in_channels=48, img_size=112)
    assert model.relu1 is False and model.conv1.weight.shape == (64, 48, 1, 1)
    prefix = models.trunk_prefix(model, n_blocks=2, avgpool_kernel=5)          # 16x16 -> floor(16/5) = 3x3 windows of 5x5
    qm = compile_brevitas_qat_model(prefix, crops[:16], n_bits=5, rounding_threshold_bits=6, p_error=0.01)
    try:
        q = qm.quantize_input(crops[16:17])
        want = _oracle(qm, q)
        assert want.shape == (1, 64 * 3 * 3)
        qm.fhe_circuit.keygen(seed=1)
        got = qm.forward_quantized(q, "execute")
        assert np.array_equal(got, want), (got, want)
    finally:
        qm.close()


def test_encrypted_config5_whole_trunk_on_a_crop_bit_exact():
    """#5 ResNet-18 48x112^2, ALL FOUR STAGES (round 2 stopped after stage 1): the 1x1 48->64 stem without ReLU (reference
    models/backbone.py:555-563), the two 64-channel blocks, then the stride-2 3x3 convolutions and 1x1 stride-2 shortcuts into 128, 256
    and 512 channels -- on a 56x56 crop of the 112x112 DCT planes (a quarter of the image: ~13 M bootstraps), pooling window 7 on the
    final 7x7 map instead of 14 on 14x14, same tiers and per-site precisions.  Every one of the 512 outputs equals the integer circuit."""
    from dctfhe import frontend, models, synthetic
    from dctfhe.quantized_module import compile_brevitas_qat_model
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=112, channels=48)
    planes = np.stack([tf(im) for im in synthetic.synthetic_images(13, 7, size=96)]).astype(np.float32)     # [13, 48, 112, 112]
    crops = planes[:, :, 28:84, 28:84]
    model = models.ResNet18QAT(bit_width=4, in_channels=48, img_size=112)
    assert len(model.blocks) == 8 and [b.C1.stride for b in model.blocks] == [1, 1, 2, 1, 2, 1, 2, 1]
    assert [b.shortcut is not None for b in model.blocks] == [False, False, True, False, True, False, True, False]
    whole = models.trunk_prefix(model, n_blocks=8, avgpool_kernel=7)
    qm = compile_brevitas_qat_model(whole, crops[:12], n_bits=5, rounding_threshold_bits=6, p_error=0.01)
    try:
        assert qm.compiled.n_out() == 512
        q = qm.quantize_input(crops[12:13])
        want = _oracle(qm, q)
        assert want.shape == (1, 512) and len(np.unique(want)) > 4
        qm.fhe_circuit.keygen(seed=1)
        got = qm.forward_quantized(q, "execute")
        assert np.array_equal(got, want), np.argwhere(got != want)
    finally:
        qm.close()


def _catalogue_variant(changes):
    import dataclasses
    from dctfhe import params as P
    ps = P.default_params()
    tiers = list(ps.tiers)
    for name, ch in changes.items():
        i = [t.name for t in tiers].index(name)
        tiers[i] = dataclasses.replace(tiers[i], lwe_sigma=0.0, **ch)
    return dataclasses.replace(ps, tiers=tiers)


@pytest.mark.parametrize("changes,expect_used,expect_unused", [
    # the one-bit tiers on their former base-8 key switch: a rounding chain then needs B, Ba AND Ba2 (the shipped catalogue has no Ba steps)
    ({"B": dict(n=584, lk=5, betak=3), "Ba": dict(n=584, lk=5, betak=3), "Ba2": dict(n=584, lk=5, betak=3)}, ("B", "Ba", "Ba2", "T4r2"), ()),
    # a two-bit refresh too noisy for the budget: the compiler falls back to the one-bit-chain refresh T4r
    ({"T4r2": dict(beta=15)}, ("T4r",), ("T4r2",)),
], ids=["chain B-Ba-Ba2", "fallback T4r"])
def test_encrypted_prefix_on_catalogue_variants_bit_exact(changes, expect_used, expect_unused):
    """paths of the engine the shipped catalogue no longer walks on the benchmark circuits (three-way rounding chains, the fallback refresh
    tier), on the full-size rings: the config-#5 prefix of the test above, encrypted == integer circuit"""
    from dctfhe import frontend, models, synthetic
    from dctfhe.quantized_module import compile_brevitas_qat_model
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=112, channels=48)
    planes = np.stack([tf(im) for im in synthetic.synthetic_images(18, 7, size=96)]).astype(np.float32)
    crops = planes[:, :, 40:56, 40:56]
    model = models.ResNet18QAT(bit_width=4, in_channels=48, img_size=112)
    prefix = models.trunk_prefix(model, n_blocks=2, avgpool_kernel=5)
    qm = compile_brevitas_qat_model(prefix, crops[:16], n_bits=5, rounding_threshold_bits=6, p_error=0.01, param_set=_catalogue_variant(changes))
    try:
        counts = qm.compiled.pbs_counts()
        assert all(counts.get(t, 0) > 0 for t in expect_used) and all(counts.get(t, 0) == 0 for t in expect_unused), counts
        q = qm.quantize_input(crops[16:17])
        want = _oracle(qm, q)
        qm.fhe_circuit.keygen(seed=2)
        got = qm.forward_quantized(q, "execute")
        assert np.array_equal(got, want), (got, want)
    finally:
        qm.close()
