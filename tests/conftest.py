import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import _oracle
    _oracle.orc()          # builds liboracle.so (and oracle/_ref when the reference tree is present)
    yield


@pytest.fixture(scope="session")
def product_lib():
    """libxz_amd.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    import subprocess
    import xz_amd
    if not os.path.exists(xz_amd.LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "xz_amd", "csrc")], check=True)
    return xz_amd.lib()
