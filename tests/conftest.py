"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`: oracle vs the reference's known-answer tests and golden fixtures, host logic,
C-ABI symbol export.  `-m gpu`: parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
