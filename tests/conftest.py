"""pytest configuration: `gpu` marker + shared fixtures.

CPU tests (-m "not gpu") cover the oracle against golden vectors / the compiled reference kernels, the
host logic and the C-ABI surface; GPU tests (-m gpu) are the parity tests proper and call through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.pyoracle import Reference
    r = Reference()
    if not r.available():
        pytest.skip("oracle/_ref/libsora_ref.so not built (reference tree absent)")
    return r


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
