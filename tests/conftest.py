import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def pytest_runtest_setup(item):
    """GPU tests: name the test on the NATIVE stderr before it starts, so that a message of the HIP / HSA runtime (which
    writes to fd 2 directly and may be the last thing a dying process says) can be attributed to a test in the log."""
    if item.get_closest_marker("gpu") is not None:
        try:
            os.write(2, f"\n[gpu test] {item.nodeid}\n".encode())
        except OSError:
            pass


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session", autouse=True)
def _built_product():
    """The product library must exist (built in-tree); build it if this checkout has none."""
    lib = ROOT / "lut_ldpc_amd" / "lib" / "liblut_ldpc_amd.so"
    if not lib.exists():
        import __graft_entry__ as ge
        ge.build()
    return lib
