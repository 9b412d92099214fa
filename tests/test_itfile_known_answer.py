"""Byte-level known answer for the .it writer: the expected file is assembled BY HAND from the reference's own MATLAB
writer / reader -- scripts/itsave.m:55 (magic 'IT++' + version byte 3), :85-99 and :134-153 (per variable: three uint64 sizes
[hdr data block], name\\0, type\\0, an empty description\\0, uint64 length, little-endian payload; hdr = 3*8 + len(name)+1 + len(type)+1 + 1)
and scripts/itload.m:89-91 ('float64': one double), :141-145 ('string': uint64 length + chars) -- and compared with what
LDPC_BER_Sim_Results::write_itfile (src/LDPC_BER_Sim.cpp:342-362: the eleven fields, counters as dvec through to_vec()) writes.
No reference-written .it file exists in the repository; this pins the container format to the scripts that read it."""
import ctypes as C
import struct

import numpy as np

from lut_ldpc_amd._capi import lib, check


def _block(name: str, typ: str, payload: bytes) -> bytes:
    hdr = 3 * 8 + len(name) + 1 + len(typ) + 1 + 1           # itsave.m:134-136
    return struct.pack("<QQQ", hdr, len(payload), hdr + len(payload)) + name.encode() + b"\0" + typ.encode() + b"\0" + b"\0" + payload


def _dvec(name, v):                                            # itsave.m:143-153
    v = np.asarray(v, "<f8").ravel()
    return _block(name, "dvec", struct.pack("<Q", len(v)) + v.tobytes())


def test_results_file_is_byte_identical_to_the_hand_assembled_one(tmp_path):
    snr = np.array([0.0, 0.5, 1.0, 4.0])
    cnt = np.array([[21, 21 * 250, 21, 1034, 1200], [100, 25000, 7, 130, 5100], [100, 25000, 0, 0, 4000], [0, 0, 0, 0, 0]], np.int64)
    path = tmp_path / "r.it"
    check(lib.lutldpc_selftest_write_results_it(str(path).encode(), snr.ctypes.data_as(C.POINTER(C.c_double)), cnt.ctypes.data_as(C.POINTER(C.c_int64)),
                                                len(snr), 500, 250, 12.625))
    want = b"IT++" + bytes([3])                                # itsave.m:55
    want += _dvec("sim_SNRdB", snr)
    for k, name in enumerate(["sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors"]):
        want += _dvec(name, cnt[:, k].astype(float))           # to_vec(ivec): the counters are stored as doubles
    want += _dvec("ldpc_nvar", [500.0]) + _dvec("ldpc_nchk", [250.0]) + _dvec("ldpc_code_rate", [0.5])
    want += _block("runtime", "float64", struct.pack("<d", 12.625))                       # itload.m:89-91
    git = b"lut_ldpc_amd-0.1"
    want += _block("gitversion", "string", struct.pack("<Q", len(git)) + git)             # itload.m:141-145
    got = path.read_bytes()
    assert got == want, (len(got), len(want), next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None))


def test_hand_assembled_bytes_follow_itsave_m_for_a_three_element_dvec():
    """The arithmetic of itsave.m spelled out once on a literal: x = [1 2.5 -3] -> hdr 31, data 32, block 63."""
    b = _dvec("x", [1.0, 2.5, -3.0])
    assert b[:24] == struct.pack("<QQQ", 3 * 8 + 2 + 5 + 1, 8 + 3 * 8, 3 * 8 + 2 + 5 + 1 + 8 + 3 * 8)
    assert b[24:32] == b"x\0dvec\0\0" and b[32:40] == struct.pack("<Q", 3) and b[40:] == struct.pack("<ddd", 1.0, 2.5, -3.0)
