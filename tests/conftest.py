import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dct-cryptonets_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ref_loader
    ref_loader.build()
    return ref_loader


@pytest.fixture(scope="session")
def gpu_ctx():
    from dctfhe.engine import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
