"""pytest configuration: registers the `gpu` marker and makes oracle/ and the package importable.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI load/export checks (no compute calls).
`-m gpu`:       parity tests proper — every call goes through the C-ABI of libggml-mi355x.so.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
