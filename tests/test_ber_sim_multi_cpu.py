"""The native multi-device ber_sim (lut_ldpc_amd/csrc/host/ber_sim_multi.cpp) without a GPU: the sanitizer build of the library on
the do-nothing HIP runtime with TWO fake devices and the fake RCCL (tests/fakehip/fake_rccl.c: the same five entry points on
host threads), two lanes per device = four ranks.  Kernels do not run there (every frame "decodes" with zero errors), so what is
checked is the host side -- threads, barriers, the frame dealing, the all-gather / all-reduce plumbing through device buffers,
handle life cycles -- under AddressSanitizer + UBSan, and that every rank configuration writes the same result file.
The arithmetic of the exchange (stop rule in global frame order) is checked with real frame statistics in
tests/test_sharded_gloo.py (same rule in Python) and on the GPU in tests/test_20_frontend_gpu.py."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(HERE))

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, {root!r})
from lut_ldpc_amd._capi import lib
argv = {argv!r}
arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
lib.lutldpc_ber_sim_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
rc = lib.lutldpc_ber_sim_main(len(argv), arr)
print("ber_sim rc", rc)
sys.exit(rc)
"""


def _basedir(tmp_path, nframes):
    base = tmp_path / "base"
    (base / "codes").mkdir(parents=True)
    (base / "results").mkdir()
    (base / "trees").mkdir()
    shutil.copy(ROOT / "data" / "codes" / "rate0.50_dv02-17_dc08-09_lut_q4_N500.alist", base / "codes")
    ini = (ROOT / "data" / "params" / "ber.ini.irregular.example").read_text()
    ini = ini.replace("Nframes  = 1e2", f"Nframes  = {nframes}\n   batch_frames = 1024")
    ini = ini.replace("LDPC.zero_codeword", "LDPC.zero_codeword")
    p = base / "ber.ini"
    p.write_text(ini)
    return base, p


def _run(base, params, extra, env_extra, tag):
    sys.path.insert(0, str(HERE / "fakehip"))
    import replay
    subprocess.run(["make", "-s", "-j8", "-C", str(HERE / "fakehip")], check=True)
    env = replay.sanitizer_env()
    env.update(env_extra)
    argv = ["ber_sim", "-p", str(params), "-b", str(base), "-s", "2", "-c", tag] + extra
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=str(ROOT), argv=argv)], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    files = sorted((base / "results").glob(f"*{tag}/*_rseed0002.it"))
    assert len(files) == 1, (list((base / "results").rglob("*")), r.stdout[-2000:])
    return files[0], r.stdout


def test_multi_device_ber_sim_on_two_fake_devices_under_asan(tmp_path):
    from itfile_reader import itload
    base, params = _basedir(tmp_path, 6000)
    fake_rccl = str(HERE / "fakehip" / "_build" / "libfakerccl.so")
    one, out1 = _run(base, params, ["-d", "0", "--lanes", "1"], {}, "_one")
    two, out2 = _run(base, params, ["-d", "0,1", "--lanes", "2", "--exchange", "rccl"], {"FAKEHIP_DEVICES": "2", "LUTLDPC_RCCL_LIB": fake_rccl}, "_rccl")
    shared, out3 = _run(base, params, ["-d", "0,0,0", "--lanes", "1", "--exchange", "host"], {}, "_host")
    assert "2 device(s) x 2 lane(s), counters over RCCL" in out2 and "3 device(s) x 1 lane(s), counters over the host" in out3
    a, b, c = itload(one), itload(two), itload(shared)
    for k in ("sim_SNRdB", "sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors", "ldpc_nvar", "ldpc_nchk"):
        assert (np.asarray(a[k]) == np.asarray(b[k])).all() and (np.asarray(a[k]) == np.asarray(c[k])).all(), k
    assert a["sim_Nframes"][0] == 6000               # (no kernel ran: no frame error, the point runs to Nframes and the sweep stops)


def test_rccl_refuses_ranks_that_share_a_device(tmp_path):
    base, params = _basedir(tmp_path, 600)
    sys.path.insert(0, str(HERE / "fakehip"))
    import replay
    env = replay.sanitizer_env()
    argv = ["ber_sim", "-p", str(params), "-b", str(base), "-d", "0,0", "--exchange", "rccl"]
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=str(ROOT), argv=argv)], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "one device per rank" in r.stderr
