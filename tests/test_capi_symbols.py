"""The C-ABI library loads on a CPU-only host and exports every function include/*.h declares."""
import ctypes
import re

from helpers import ROOT


def _declared(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    return sorted(set(re.findall(r"\b(lutldpc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(str(ROOT / "lut_ldpc_amd" / "lib" / "liblut_ldpc_amd.so"))
    names = []
    for h in sorted((ROOT / "include").glob("*.h")):
        names += _declared(h.read_text())
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_device_count_without_gpu():
    import lut_ldpc_amd as L
    from lut_ldpc_amd._capi import lib
    assert b"gfx950" in lib.lutldpc_version()
    assert L.device_count() >= 0


def test_shipped_library_is_newer_than_its_sources():
    """lut_ldpc_amd/lib/ is git-ignored and travels to the GPU box as a prebuilt binary: a stale build must not ship."""
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    lib = root / "lut_ldpc_amd" / "lib" / "liblut_ldpc_amd.so"
    srcs = [p for p in (root / "lut_ldpc_amd" / "csrc").rglob("*") if p.suffix in (".hip", ".hpp", ".cpp", ".h")] + list((root / "include").glob("*.h"))
    assert srcs
    newer = [str(p.relative_to(root)) for p in srcs if p.stat().st_mtime > lib.stat().st_mtime]
    assert not newer, f"rebuild (make -C lut_ldpc_amd/csrc): newer than the library: {newer}"
