"""Thin Python handle over the [BP] comparison decoder (include/lut_ldpc_bp.h): itpp::LDPC_Code::bp_decode for a batch of
frames on the MI355X.  PARITY UNPINNED against the reference's forked IT++ (absent); the header states the arithmetic."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._capi import lib, check

_vp, _ip, _u8p, _dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
for _name, (_res, _args) in {
    "lutldpc_bp_create": (C.c_int, [C.c_int, C.c_int, _ip, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lutldpc_bp_destroy": (C.c_int, [_vp]),
    "lutldpc_bp_set_exit_conditions": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "lutldpc_bp_logexp_table": (C.c_int, [_vp, _ip, C.c_int]),
    "lutldpc_bp_decode_llr_batch": (C.c_int, [_vp, _dp, C.c_int, _u8p, _ip, _ip]),
    "lutldpc_bp_decode_qllr_batch": (C.c_int, [_vp, _ip, C.c_int, _u8p, _ip, _ip]),
    "lutldpc_awgn_llr": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_double, _u8p, _dp, _ip]),
}.items():
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class BPDecoder:
    """LLR_calc_unit(d1, d2, d3, d4) + LDPC_Code::bp_decode (src/LDPC_BER_Sim.cpp:199-200): defaults as the reference's
    [BP] section (qllr_scale_res 12, qllr_table_size 300, qllr_spacing_res 7, qllr_total_res 28)."""

    def __init__(self, nvar, nchk, dv, dc, cn_msg_idx, d1=12, d2=300, d3=7, d4=28, device=0):
        dv, dc, cn = (np.ascontiguousarray(a, np.int32) for a in (dv, dc, cn_msg_idx))
        self.nvar, self.nchk = int(nvar), int(nchk)
        self._h = _vp()
        check(lib.lutldpc_bp_create(self.nvar, self.nchk, _p(dv, C.c_int32), _p(dc, C.c_int32), _p(cn, C.c_int32), d1, d2, d3, d4, int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib.lutldpc_bp_destroy(self._h)
            self._h = None

    __del__ = close

    def set_exit_conditions(self, max_iters, psc=True, pisc=False):
        check(lib.lutldpc_bp_set_exit_conditions(self._h, int(max_iters), int(psc), int(pisc)))

    def logexp_table(self) -> np.ndarray:
        n = lib.lutldpc_bp_logexp_table(self._h, None, 0)
        out = np.zeros(max(n, 1), np.int32)
        lib.lutldpc_bp_logexp_table(self._h, _p(out, C.c_int32), n)
        return out[:n]

    def _decode(self, fn, a, t, want_qllr):
        B, N = a.shape
        if N != self.nvar:
            raise ValueError("input must be [B, nvar]")
        bits, iters = np.empty((B, N), np.uint8), np.empty(B, np.int32)
        q = np.empty((B, N), np.int32) if want_qllr else None
        check(fn(self._h, _p(a, t), B, _p(bits, C.c_uint8), _p(iters, C.c_int32), _p(q, C.c_int32) if want_qllr else None))
        return (bits, iters, q) if want_qllr else (bits, iters)

    def decode_llr_batch(self, llr, want_qllr=False):
        return self._decode(lib.lutldpc_bp_decode_llr_batch, np.ascontiguousarray(llr, np.float64), C.c_double, want_qllr)

    def decode_qllr_batch(self, qllr, want_qllr=False):
        return self._decode(lib.lutldpc_bp_decode_qllr_batch, np.ascontiguousarray(qllr, np.int32), C.c_int32, want_qllr)


def awgn_llr(seed, stream, frame0, B, N, N0, codewords=None):
    """The [BP] path's host front end (include/lut_ldpc_host.h: lutldpc_awgn_llr): (llr [B, N] float64, uncoded errors [B])."""
    llr, unc = np.empty((B, N), np.float64), np.empty(B, np.int32)
    cw = None if codewords is None else np.ascontiguousarray(codewords, np.uint8)
    check(lib.lutldpc_awgn_llr(int(seed), int(stream), int(frame0), int(B), int(N), float(N0), _p(cw, C.c_uint8) if cw is not None else None,
                               _p(llr, C.c_double), _p(unc, C.c_int32)))
    return llr, unc
