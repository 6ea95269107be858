"""The CLI mirror keeps every flag name / default / choice of the reference's io_utils.parse_args('homomorphic_eval')
(reference io_utils.py:19-45, 69-85)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_FLAGS = {          # name: default   (io_utils.py)
    "dataset": "cifar10", "model": "ResNet18qat", "num_classes": 10, "dataset_path": None, "save_path": None, "train_aug": False,
    "dct_status": False, "channels": 64, "filter_size": 8, "image_size": 32, "image_size_dct": 56, "dct_pattern": "default", "bit_width": 4,
    "dropout": None, "verbose": True, "checkpoint_path": None, "calib_batch_size": 64, "test_batch_size": 1, "test_subset": 1,
    "fhe_mode": "simulate", "rounding_threshold_bits": 6, "n_bits": 5, "p_error": 0.01, "reliability_test": True,
}


def test_flag_names_and_defaults(monkeypatch):
    spec = importlib.util.spec_from_file_location("he_cli", os.path.join(ROOT, "dct-cryptonets_amd", "homomorphic_eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["homomorphic_eval.py"])
    ns = vars(mod.parse_args())
    for k, v in REFERENCE_FLAGS.items():
        assert k in ns and ns[k] == v, (k, ns.get(k), v)
    monkeypatch.setattr(sys, "argv", ["homomorphic_eval.py", "--model", "ResNet20qat", "--dct_status", "--channels", "24", "--fhe_mode", "execute"])
    ns = vars(mod.parse_args())
    assert ns["model"] == "ResNet20qat" and ns["dct_status"] is True and ns["channels"] == 24 and ns["fhe_mode"] == "execute"
