"""Minimal reader of IT++ .it (version 3) files, following scripts/itload.m:48-63 of the reference."""
import struct

import numpy as np


def itload(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"IT++" and raw[4] == 3, "not an IT++ v3 file"
    pos, out = 5, {}
    while pos + 24 <= len(raw):
        hdr, dat, tot = struct.unpack_from("<QQQ", raw, pos)
        name_end = raw.index(b"\0", pos + 24)
        name = raw[pos + 24:name_end].decode()
        type_end = raw.index(b"\0", name_end + 1)
        typ = raw[name_end + 1:type_end].decode()
        d = raw[pos + hdr:pos + hdr + dat]
        if typ == "bin":
            out[name] = d[0]
        elif typ == "int32":
            out[name] = struct.unpack("<i", d)[0]
        elif typ == "float64":
            out[name] = struct.unpack("<d", d)[0]
        elif typ in ("dvec", "ivec", "bvec", "string"):
            n = struct.unpack_from("<Q", d)[0]
            dt = {"dvec": "<f8", "ivec": "<i4", "bvec": "u1", "string": "S1"}[typ]
            arr = np.frombuffer(d, dt, n, 8)
            out[name] = arr.tobytes().decode() if typ == "string" else arr.copy()
        elif typ == "ivecArray":
            n = struct.unpack_from("<Q", d)[0]
            p, a = 8, []
            for _ in range(n):
                m = struct.unpack_from("<Q", d, p)[0]
                a.append(np.frombuffer(d, "<i4", m, p + 8).copy())
                p += 8 + 4 * m
            out[name] = a
        pos += tot
    return out
