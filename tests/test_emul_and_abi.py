"""(1) host emulation of the device thread programs (FFT index math, whole PBS) against the oracle;
(2) the C-ABI library loads and exports every symbol include/dctfhe.h declares; no compute without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_thread_programs(oracle):
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(d, "emul_pbs")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "EMUL OK" in out.stdout, out.stdout + out.stderr


def test_abi_exports_every_declared_symbol():
    from dctfhe import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    hdr = open(os.path.join(ROOT, "include", "dctfhe.h")).read()
    declared = sorted(set(re.findall(r"\b(dctfhe_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_lib.EXPORTS) == declared


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dctfhe._lib import DctfheError
    from dctfhe.engine import Context
    with pytest.raises(DctfheError, match="no CPU path"):
        Context(0)


def test_build_recorded_kernel_resources():
    """__graft_entry__.build() records registers / scratch / occupancy of every kernel it compiles and refuses a library whose bootstrap
    kernels fell off the register cliff (hipcc has demoted their register arrays to scratch over one-line edits: DESIGN.md section 5).
    Here: the record of the library in the tree, when this checkout built it."""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dct-cryptonets_amd", "build_resources.txt")
    if not os.path.exists(path):
        import pytest
        pytest.skip("library not built by this checkout's build() (no record)")
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("#")]
    pbs = [r for r in rows if "pbs_kernel" in r[0]]
    assert len(pbs) >= 20
    for name, vgpr, agpr, scratch, occ in pbs:
        p16 = re.search(r"pbs_kernelILi\d+ELi\d+ELi\d+ELi16E", name) is not None
        assert int(scratch) <= 128 and (int(occ) >= 2 or p16), (name, vgpr, scratch, occ)
