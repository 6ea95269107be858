"""P1: decrypt(run(encrypt(q))) == noise-free integer circuit, element for element.
Small model + small (insecure) rings so the whole thing also fits the CPU oracle; the full-size
ResNet-20 run lives in tests/test_gpu_resnet20.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from dctfhe import models, params as P
    from dctfhe.quantized_module import compile_brevitas_qat_model
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    qm = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits=6, param_set=P.test_params())
    yield qm, calib
    qm.close()


def _oracle_out(qm, q):
    from oracle import circuit_ref
    out, ov = circuit_ref.run_clear(qm.compiled.blob, qm.encode_input(q))
    assert not ov
    return qm.decode_output(out)


def test_clear_mode_matches_oracle(tiny):
    qm, calib = tiny
    q = qm.quantize_input(calib[:8])
    assert np.array_equal(qm.forward_quantized(q, "disable"), _oracle_out(qm, q))


def test_execute_matches_integer_circuit(tiny):
    qm, calib = tiny
    qm.fhe_circuit.keygen(seed=5)
    q = qm.quantize_input(calib[:6])
    want = _oracle_out(qm, q)
    got = qm.forward_quantized(q, "execute")
    assert got.shape == want.shape == (6, qm.compiled.n_out())
    assert np.array_equal(got, want)
    # the float surface: forward() == dequantised integers
    assert np.allclose(qm.forward(calib[:6], fhe="execute"), want * qm.compiled.out_scale)
    assert qm.fhe_circuit.graph.maximum_integer_bit_width() <= 16
    assert "round_lut" in qm.fhe_circuit.mlir


def test_session_runs_twice_on_one_upload(tiny):
    """dctfhe_session_run keeps its input: upload once, run twice (what bench.py does for warm-up + steps), both passes
    equal the integer circuit (the allocator used to recycle the input buffer)."""
    qm, calib = tiny
    qm.fhe_circuit.keygen(seed=5)
    q = qm.quantize_input(calib[20:23])
    want = _oracle_out(qm, q)
    sess = qm._session("execute", 3)
    sess.upload(qm._keys.encrypt(qm.encode_input(q).reshape(-1)))
    for _ in range(2):
        sess.run()
        out = sess.download().reshape(-1, qm._keys.D + 1)
        assert np.array_equal(qm.decode_output(qm._keys.decrypt(out).reshape(3, -1)), want)


def test_ragged_batches(tiny):
    """batch sizes that do not fill the last workgroup / chunk"""
    qm, calib = tiny
    for B in (1, 3):
        q = qm.quantize_input(calib[10:10 + B])
        assert np.array_equal(qm.forward_quantized(q, "execute"), _oracle_out(qm, q))


def test_cli_mirror_simulate_mode():
    """the homomorphic_eval.py-compatible driver with the reference's flags (run_homomorphic_eval.sh ResNet20 CIFAR block)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "dct-cryptonets_amd", "homomorphic_eval.py"), "--dataset", "cifar10", "--model", "ResNet20qat",
           "--dct_status", "--channels", "24", "--filter_size", "4", "--image_size_dct", "16", "--bit_width", "4", "--fhe_mode", "simulate",
           "--calib_batch_size", "32", "--test_batch_size", "2", "--test_subset", "4", "--rounding_threshold_bits", "6", "--n_bits", "5",
           "--p_error", "0.01"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(os.environ.get("TMPDIR", "/tmp")))
    assert out.returncode == 0, out.stderr[-2000:]
    for needle in ("Time for FHE compilation", "Max bit-width:", "it works in FHE!!", "Keygen time:", "Time per inference in FHE",
                   "Encrypted Reliability Analysis Results", "Encrypted top1 acc:", "Done"):
        assert needle in out.stdout, out.stdout
    # the sweep of the reference (random states 27, 28): encrypted == unencrypted on every subset at the exact tiers
    import re
    plain = re.search(r"Unencrypted top1 acc: (.*)", out.stdout).group(1)
    enc = re.search(r"Encrypted top1 acc: (.*)", out.stdout).group(1)
    assert plain == enc and plain.count(",") == 1


def test_approximate_rounding_small_rings():
    """rounding_threshold_bits={"n_bits": 6, "method": "approximate"} (reference README.md:95-114): no one-bit steps, the
    table bootstrap rounds.  The mod-switch noise alone exceeds the half-unit margin of the inputs next to a rounding
    boundary, so those land on either neighbour: most outputs equal the integer circuit, the rest are a unit or two off."""
    from dctfhe import models, params as P
    from dctfhe.quantized_module import compile_brevitas_qat_model
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    qm = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits={"n_bits": 6, "method": "approximate"},
                                    param_set=P.test_params())
    try:
        assert qm.compiled.rounding_method == "approximate" and "rounding=approximate" in qm.fhe_circuit.mlir
        assert set(qm.compiled.pbs_counts()) == {"t"}                      # table bootstraps only
        qm.fhe_circuit.keygen(seed=6)
        q = qm.quantize_input(calib[:5])
        diff = np.abs(qm.forward_quantized(q, "execute") - _oracle_out(qm, q))
        print("approximate rounding: exact outputs %.1f%%, max |diff| %d" % (100.0 * (diff == 0).mean(), diff.max()))
        assert (diff == 0).mean() > 0.5 and diff.max() <= 4
    finally:
        qm.close()


def test_p_error_tier_policy_full_size_rings():
    """tier_policy="p_error" + approximate rounding on the small model with the full-size p_error catalogue:
    stochastic by design -- most outputs equal the integer circuit, the rest are a few units off (boundary flips of the
    6-bit rounding propagate through the residual blocks)."""
    from dctfhe import models
    from dctfhe.quantized_module import compile_brevitas_qat_model
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    qm = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits={"n_bits": 6, "method": "approximate"},
                                    p_error=0.01, tier_policy="p_error")
    try:
        names = set(qm.compiled.pbs_counts())
        assert names <= {"F6", "F5", "T4", "T5a"} and "F6" in names
        assert max(o.pfail for o in qm.compiled.ops if o.type == 4) <= 0.01
        qm.fhe_circuit.keygen(seed=8)
        q = qm.quantize_input(calib[:4])
        got, want = qm.forward_quantized(q, "execute"), _oracle_out(qm, q)
        diff = np.abs(got - want)
        print("p_error policy: exact outputs %.1f%%, max |diff| %d" % (100.0 * (diff == 0).mean(), diff.max()))
        assert (diff == 0).mean() > 0.5 and diff.max() <= 4
    finally:
        qm.close()


def test_simulate_samples_the_noise_model():
    """fhe="simulate" = the integer circuit with the compiler's noise model sampled at every look-up: at the
    exact tiers it coincides with the clear circuit; with approximate rounding and the p_error tiers it shows the same kind
    of deviation as the encrypted run (boundary flips), and two simulations draw different noise."""
    from dctfhe import models, params as P
    from dctfhe.quantized_module import compile_brevitas_qat_model
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    exact = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits=6)      # default (exact) catalogue
    try:
        q = exact.quantize_input(calib[:8])
        assert np.array_equal(exact.forward_quantized(q, "simulate"), exact.forward_quantized(q, "disable"))
    finally:
        exact.close()
    fast = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits={"n_bits": 6, "method": "approximate"},
                                      p_error=0.01, tier_policy="p_error")
    try:
        q = fast.quantize_input(calib[:8])
        clear = fast.forward_quantized(q, "disable")
        s1, s2 = fast.forward_quantized(q, "simulate"), fast.forward_quantized(q, "simulate")
        d1 = np.abs(s1 - clear)
        assert 0 < (d1 != 0).mean() < 0.6 and d1.max() <= 4          # deviates, mildly
        assert not np.array_equal(s1, s2)                           # fresh draws every run
        assert np.array_equal(fast.forward_quantized(q, "disable"), clear)
    finally:
        fast.close()


def test_input_on_a_key_prefix_and_tail_guard():
    """dctfhe_params.input_dim: the client masks only a prefix of the big key, the compiler propagates the effective
    dimension and the key switch / convolutions skip the zero tail -- same outputs as the integer circuit; ciphertexts that
    are not zero beyond that prefix are refused at upload (they would be evaluated wrongly)."""
    import dataclasses
    from dctfhe import _lib, models, params as P
    from dctfhe.quantized_module import compile_brevitas_qat_model
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    ps = dataclasses.replace(P.test_params(), input_dim=512)
    qm = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits=6, param_set=ps)
    try:
        assert qm.compiled.tensors[qm.compiled.input_tensor].deff == 512
        qm.fhe_circuit.keygen(seed=11)
        q = qm.quantize_input(calib[:3])
        assert np.array_equal(qm.forward_quantized(q, "execute"), _oracle_out(qm, q))
        cts = qm._keys.encrypt(qm.encode_input(q).reshape(-1)).reshape(-1, qm._keys.D + 1)
        assert not cts[:, 512:qm._keys.D].any()
        cts[5, 700] = 1
        sess = qm._session("execute", 3)
        with pytest.raises(_lib.DctfheError, match="beyond 512"):
            sess.upload(cts)
        # the compact wire form (rows of dim mask words + body): the same guard on rows wider than the effective dimension, rows of
        # exactly input_dim + 1 words accepted, and a row width the key does not have refused
        wide = qm._keys.encrypt(qm.encode_input(q).reshape(-1), 640)
        assert wide.shape[1] == 641 and not wide[:, 512:640].any()
        sess.upload(wide, 640)
        wide[7, 600] = 3
        with pytest.raises(_lib.DctfheError, match="beyond 512"):
            sess.upload(wide, 640)
        sess.upload(qm._keys.encrypt(qm.encode_input(q).reshape(-1), 512), 512)
        sess.run()
        out_dim = sess.dims()[1]
        got = qm.decode_output(qm._keys.decrypt(sess.download(out_dim).reshape(-1, out_dim + 1), out_dim).reshape(3, -1))
        assert np.array_equal(got, _oracle_out(qm, q))
        with pytest.raises(_lib.DctfheError, match="mask words"):
            sess.upload(np.zeros((wide.shape[0], qm._keys.D + 65), np.uint64), qm._keys.D + 64)
        with pytest.raises(_lib.DctfheError, match="mask words"):
            qm._keys.encrypt(qm.encode_input(q).reshape(-1), 256)            # narrower than what these parameters mask
        with pytest.raises(_lib.DctfheError, match="the output needs"):
            sess.download(out_dim - 8)
    finally:
        qm.close()


def test_imported_checkpoint_runs_encrypted(tmp_path):
    """SURVEY 8f-1 end to end: a `best.tar` in the reference's layout (train.py:82-89; DataParallel + `feature.trunk.<i>` keys,
    Brevitas activation thresholds included) -> load_checkpoint -> compile with the imported scales -> encrypted run == the
    integer circuit."""
    import torch
    from dctfhe import checkpoint, models, params as P
    from dctfhe.quantized_module import compile_brevitas_qat_model
    from test_checkpoint import _fake_state
    rng = np.random.default_rng(4)
    model = models.tiny_resnet_q()
    path = str(tmp_path / "best.tar")
    torch.save({"epoch": 1, "state": _fake_state(model, rng), "prec1": 50.0, "prec5": 90.0, "optimizer": {}}, path)
    meta, unused = checkpoint.load_checkpoint(path, model)
    assert meta["epoch"] == 1 and all("num_batches_tracked" in k for k in unused)
    assert len(model.act_scales) == 4 + 3 * len(model.blocks) + 1      # stem x3 + final, 3 per block, + the one 1x1-shortcut block's BNquant_out
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    qm = compile_brevitas_qat_model(model, calib, n_bits=5, rounding_threshold_bits=6, param_set=P.test_params())
    try:
        assert abs(qm.compiled.in_scale - 2.0 / 8) < 1e-7          # the checkpoint's quant_inp threshold, not a re-calibrated one
        qm.fhe_circuit.keygen(seed=12)
        q = qm.quantize_input(calib[:4])
        assert np.array_equal(qm.forward_quantized(q, "execute"), _oracle_out(qm, q))
    finally:
        qm.close()


def test_rounding_chain_on_a_key_switch_that_cannot_narrow():
    """a rounding chain works in place on a shifted copy of its input of which only the first `deff` mask words used to be
    written; a key switch that cannot narrow to `deff` (betak = 8 runs the integer-VALU GEMM, which walks the whole row) then read
    whatever the recycled buffer held beyond them.  Here: table ring = D = 1024, client encryption on a 512-word prefix (input_dim),
    bit tier ring 512 -- so every work row has 512 meaningful mask words in 1024 -- and betak = 8 on both tiers.  Two passes over one
    upload (the second finds the first's leftovers in every buffer), all outputs equal to the integer circuit."""
    from dctfhe import models, params as P
    from dctfhe.quantized_module import compile_brevitas_qat_model
    t_tab = P.TierSpec("t", n=48, k=1, logN=10, l=2, beta=12, lk=3, betak=8, lwe_sigma=2.0 ** -28, glwe_sigma=2.0 ** -48)
    t_bit = P.TierSpec("b", n=40, k=2, logN=8, l=2, beta=10, lk=3, betak=8, lwe_sigma=2.0 ** -26, glwe_sigma=2.0 ** -48)
    ps = P.ParamSet(D=1024, tiers=[t_tab, t_bit], bit_tier=1, table_tier_for_w={6: 0}, input_sigma=2.0 ** -55, input_dim=512)
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    qm = compile_brevitas_qat_model(models.tiny_resnet_q(), calib, n_bits=5, rounding_threshold_bits=6, param_set=ps)
    try:
        assert any(o.type == 4 and o.r > 0 for o in qm.compiled.ops)            # rounding chains are what this is about
        qm.fhe_circuit.keygen(seed=8)
        q = qm.quantize_input(calib[:4])
        want = _oracle_out(qm, q)
        sess = qm._session("execute", 4)
        in_dim, out_dim = sess.dims()
        assert in_dim == out_dim == 1024                                        # a VALU-path tier keeps every tensor at full width
        sess.upload(qm._keys.encrypt(qm.encode_input(q).reshape(-1), 512), 512)
        for _ in range(2):
            sess.run()
            out = sess.download(out_dim).reshape(-1, out_dim + 1)
            assert np.array_equal(qm.decode_output(qm._keys.decrypt(out, out_dim).reshape(4, -1)), want)
        assert np.array_equal(qm.forward_quantized(q, "execute"), want)
    finally:
        qm.close()


def test_fhe_circuit_statistics_is_a_property(tiny):
    """Concrete's `fhe_circuit.statistics` is a property (a paste had turned it into a plain method and left dead copies of
    the owner's key methods on FHECircuit)"""
    from dctfhe.quantized_module import FHECircuit
    qm, _ = tiny
    assert isinstance(FHECircuit.statistics, property)
    st = qm.fhe_circuit.statistics
    assert st.n_ops == len(qm.compiled.ops) and st.lut_sites > 0 and st.max_bit_width == qm.compiled.max_bit_width
    assert callable(qm.fhe_circuit.export_evaluation_keys) and callable(qm.fhe_circuit.load_evaluation_keys)
