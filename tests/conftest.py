import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "python-zlib-ng_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fastq():
    """3 578 369 bytes of FASTQ text: payload of the reference's tests/data/test.fastq.gz."""
    with gzip.open(os.path.join(GOLDEN, "test.fastq.gz"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def ctx():
    from zlib_ng_amd import _lib
    return _lib.default_context()
