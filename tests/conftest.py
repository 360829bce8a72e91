import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test timeout (pytest-timeout)")


def pytest_sessionstart(session):
    """Built artefacts are kept out of git; (re)build them when a fresh checkout runs the tests."""
    lib = os.path.join(ROOT, "dcreg_amd", "lib", "libdcreg_hip.so")
    runner = os.path.join(ROOT, "dcreg_amd", "bin", "icp_test_runner")
    oracle = os.path.join(ROOT, "oracle", "libdcreg_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(runner) and os.path.exists(oracle)):
        import __graft_entry__ as g
        g.build()
