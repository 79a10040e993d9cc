import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(path, *targets):
    subprocess.run(["make", "-C", path, *targets], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session", autouse=True)
def _built_test_libs():
    """Build the CPU-side checkers (make: a no-op when they are newer than their sources; a stale libemul.so once let
    an edited vcm_core.h pass untested).  The HIP library is built by __graft_entry__.build(); tests never rebuild it."""
    _make(os.path.join(ROOT, "oracle"), "liboracle.so")
    _make(os.path.join(ROOT, "tests", "host_emul"))
    yield
