"""Thin Python handle over the HIP decoder C-ABI (include/lut_ldpc_hip.h).

`Decoder` corresponds to the decode half of the reference's LDPC_Code_LUT
(src/LDPC_Code_LUT.hpp:66-366): graph index arrays + LUT trees in, `lut_decode` /
`decode(llr)` for a batch of frames out.  All arithmetic runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import json

import numpy as np

from . import _capi
from ._capi import lib, check


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Decoder:
    def __init__(self, nvar, nchk, dv, dc, cn_msg_idx, nq_cha, nq_msg, reuse_vec, max_iters, min_lut,
                 var_trees_txt, chk_trees_txt="", device=0):
        self.nvar, self.nchk = int(nvar), int(nchk)
        self.max_iters = int(max_iters)
        dv, dc, cn = _i32(dv), _i32(dc), _i32(cn_msg_idx)
        nq, ru = _i32(nq_msg), np.ascontiguousarray(reuse_vec, np.uint8)
        assert len(nq) == max_iters and len(ru) == max_iters
        self._h = C.c_void_p()
        self._owned = True
        check(lib.lutldpc_decoder_create(self.nvar, self.nchk, _p(dv, C.c_int32), _p(dc, C.c_int32), _p(cn, C.c_int32),
                                         int(nq_cha), _p(nq, C.c_int32), _p(ru, C.c_uint8), self.max_iters, int(bool(min_lut)),
                                         var_trees_txt.encode(), (chk_trees_txt or "").encode(), int(device),
                                         C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and getattr(self, "_owned", False):
            lib.lutldpc_decoder_destroy(self._h)
        self._h = None

    def __del__(self):
        self.close()

    def set_exit_conditions(self, max_iters, psc=True, pisc=False):
        check(lib.lutldpc_decoder_set_exit_conditions(self._h, int(max_iters), int(psc), int(pisc)))
        self.max_iters = int(max_iters)

    # ---- host buffers ---------------------------------------------------------------------
    def lut_decode_batch(self, cha: np.ndarray, msg0: np.ndarray):
        cha = np.ascontiguousarray(cha, np.uint8)
        msg0 = np.ascontiguousarray(msg0, np.uint8)
        B, N = cha.shape
        if N != self.nvar or msg0.shape != cha.shape:
            raise ValueError("cha/msg0 must be [B, nvar]")
        out = np.empty((B, N), np.uint8)
        iters = np.empty(B, np.int32)
        check(lib.lutldpc_decoder_decode_batch(self._h, _p(cha, C.c_uint8), _p(msg0, C.c_uint8), B, _p(out, C.c_uint8),
                                               _p(iters, C.c_int32)))
        return out, iters

    def lut_decode_batch_trace(self, cha, msg0, level, n_edges):
        """Debug path: (bits, iters, trace uint8 [n_dumps, B, E]) -- the edge messages after the initialisation, (level 3) every
        check pass and every variable pass, in the print order of output_verbosity = level (src/LDPC_Code_LUT.cpp:292-337)."""
        cha = np.ascontiguousarray(cha, np.uint8); msg0 = np.ascontiguousarray(msg0, np.uint8)
        B, N = cha.shape
        n_dumps = 1 + self.max_iters * (int(level) - 1)
        out = np.empty((B, N), np.uint8); iters = np.empty(B, np.int32)
        trace = np.empty((n_dumps, B, int(n_edges)), np.uint8)
        got = C.c_int32()
        check(lib.lutldpc_decoder_decode_batch_trace(self._h, _p(cha, C.c_uint8), _p(msg0, C.c_uint8), B, int(level), _p(out, C.c_uint8), _p(iters, C.c_int32),
                                                     _p(trace, C.c_uint8), trace.size, C.byref(got)))
        assert got.value == n_dumps
        return out, iters, trace

    def decode_llr_batch(self, llr, qb_cha, qb_msg, mode=0, cha2msg_map=None):
        llr = np.ascontiguousarray(llr, np.float64)
        B, N = llr.shape
        qc = np.ascontiguousarray(qb_cha, np.float64)
        qm = np.ascontiguousarray(qb_msg if qb_msg is not None else [], np.float64)
        mp = _i32(cha2msg_map) if cha2msg_map is not None else None
        out = np.empty((B, N), np.uint8)
        iters = np.empty(B, np.int32)
        check(lib.lutldpc_decoder_decode_llr_batch(self._h, _p(llr, C.c_double), B, _p(qc, C.c_double), len(qc),
                                                   _p(qm, C.c_double), len(qm), int(mode),
                                                   _p(mp, C.c_int32) if mp is not None else None,
                                                   _p(out, C.c_uint8), _p(iters, C.c_int32)))
        return out, iters

    # ---- device buffers (raw pointers, e.g. torch.Tensor.data_ptr()) -------------------------
    def lut_decode_batch_device(self, d_cha: int, d_msg0: int, B: int, d_out_bits: int, d_out_iters: int, sync=False):
        check(lib.lutldpc_decoder_decode_batch_device(self._h, C.c_void_p(d_cha), C.c_void_p(d_msg0), int(B),
                                                      C.c_void_p(d_out_bits), C.c_void_p(d_out_iters), int(sync)))

    @property
    def stream(self) -> int:
        return lib.lutldpc_decoder_stream(self._h) or 0

    # ---- measurement -----------------------------------------------------------------------
    def set_profiling(self, on: bool):
        check(lib.lutldpc_decoder_set_profiling(self._h, int(on)))

    def reset_profile(self):
        check(lib.lutldpc_decoder_reset_profile(self._h))

    def profile(self) -> dict:
        out = {}
        for k, name in enumerate(_capi.KIND_NAMES):
            ms, n = C.c_double(), C.c_int64()
            check(lib.lutldpc_decoder_get_profile(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = {"ms": ms.value, "launches": n.value}
        return out

    def device_bytes(self) -> int:
        return lib.lutldpc_decoder_device_bytes(self._h)

    def describe(self) -> dict:
        return json.loads(lib.lutldpc_decoder_describe(self._h).decode())

    def jit_source(self, kind: int, tree_set: int, cls: int, compile: bool = False) -> str:
        """HIP source generated for a variable (0) / decision (2) class; compile=True also runs hiprtc (no GPU needed)."""
        n = lib.lutldpc_selftest_jit_source(self._h, kind, tree_set, cls, None, 0, int(compile))
        if n < 0:
            check(int(n))
        buf = C.create_string_buffer(n)
        lib.lutldpc_selftest_jit_source(self._h, kind, tree_set, cls, buf, n, 0)
        return buf.value.decode()

    def resident_source(self, G: int, compile: bool = False):
        """HIP source of the LDS-resident decode kernel for G frame groups, and (sets per workgroup, threads, LDS bytes)."""
        info = (C.c_int32 * 3)()
        n = lib.lutldpc_selftest_resident_source(self._h, int(G), None, 0, int(compile), info)
        if n < 0:
            check(int(n))
        buf = C.create_string_buffer(n)
        lib.lutldpc_selftest_resident_source(self._h, int(G), buf, n, 0, info)
        return buf.value.decode(), tuple(info)

    # ---- compile-step self test (host only) ----------------------------------------------------
    def program_eval(self, kind: int, tree_set: int, cls: int, inputs, n_out: int):
        a = _i32(inputs)
        out = np.zeros(n_out, np.int32)
        check(lib.lutldpc_selftest_program_eval(self._h, kind, tree_set, cls, _p(a, C.c_int32), len(a), _p(out, C.c_int32), n_out))
        return out

    def program_stats(self, kind: int, tree_set: int, cls: int):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.lutldpc_selftest_program_stats(self._h, kind, tree_set, cls, C.byref(a), C.byref(b), C.byref(c)))
        return {"ops": a.value, "ops_naive": b.value, "slots": c.value}
