"""The HOST half of the library (everything in lut_ldpc_amd/csrc except the device code) under AddressSanitizer + UBSan:
a sanitizer build linked against a do-nothing HIP runtime (tests/fakehip/) replays the call sequences of the GPU tests --
buffer sizing, copies, role and item tables, graph capture, handle life cycle -- with "device" memory on the ASan heap.
A heap overflow, a copy past an allocation, a leak of device buffers or undefined behaviour in the host code fails here,
on the CPU, instead of as a layout-dependent abort on the GPU box."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


def _replay(which, timeout):
    r = subprocess.run([sys.executable, str(HERE / "fakehip" / "replay.py"), "--launch", which], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-6000:])
    assert "replay ok" in r.stdout and "still live 0" in r.stdout, r.stdout[-2000:]


def test_host_half_under_asan_quick():
    _replay("quick", 900)


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("LUTLDPC_SLOW_TESTS") != "1", reason="several minutes: set LUTLDPC_SLOW_TESTS=1")
@pytest.mark.parametrize("which", ["parity", "skew", "big"])
def test_host_half_under_asan_full(which):
    _replay(which, 3000)
