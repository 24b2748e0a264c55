import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand from oracle/ with gcc."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def exact():
    """The binary128 arbiter (oracle/poolgen_exact.c; test infrastructure)."""
    import oracle_lib
    return oracle_lib.load_exact()


@pytest.fixture(scope="session")
def native():
    from poolgen_amd import load_library
    return load_library()


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from poolgen_amd import Engine
    return Engine(0)
