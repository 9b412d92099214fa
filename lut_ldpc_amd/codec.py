"""Python view of the C++ host mirror of LDPC_Code_LUT (include/lut_ldpc_host.h).

`Codec` = parity-check matrix + (optional) systematic generator + LDPC_Code_LUT, i.e. what
LDPC_BER_Sim_LUT::load builds in the reference (src/LDPC_BER_Sim.cpp:434-550).  LUT design runs in
the C++ host code, decoding in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._capi import lib, check
from .decoder import Decoder


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Codec:
    def __init__(self, alist_path=None, with_generator=False, known_rank=0, device=0, codec_path=None):
        self._h = C.c_void_p()
        if codec_path is not None:
            check(lib.lutldpc_codec_load(str(codec_path).encode(), int(device), C.byref(self._h)))
        else:
            check(lib.lutldpc_codec_create(str(alist_path).encode(), int(with_generator), int(known_rank), int(device), C.byref(self._h)))
        n, m, e, r = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.lutldpc_codec_dims(self._h, n, m, e, r))
        self.nvar, self.nchk, self.nedges, self.rank = n.value, m.value, e.value, r.value
        self.ninfo = self.nvar - self.rank
        self.max_iters = 0

    def close(self):
        if getattr(self, "_h", None):
            lib.lutldpc_codec_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- set-up -------------------------------------------------------------------------------
    def design_luts(self, tree_method="auto_bin_balanced", min_lut=True, sigma2=0.88 ** 2, max_iters=50, reuse_vec=None,
                    nq_cha=16, nq_msg=16, allow_degree_one=False) -> float:
        reuse = np.zeros(max_iters, np.uint8) if reuse_vec is None else np.ascontiguousarray(reuse_vec, np.uint8)
        nq = np.full(max_iters, nq_msg, np.int32) if np.isscalar(nq_msg) else np.ascontiguousarray(nq_msg, np.int32)
        sig = C.c_double()
        check(lib.lutldpc_codec_design_luts(self._h, tree_method.encode(), int(min_lut), float(sigma2), int(max_iters),
                                            _p(reuse, C.c_uint8), int(nq_cha), _p(nq, C.c_int32), int(allow_degree_one), C.byref(sig)))
        self.max_iters = max_iters
        self.design_from_cache = bool(lib.lutldpc_codec_design_from_cache(self._h))   # LUTLDPC_DESIGN_CACHE=<dir>
        return sig.value

    def set_exit_conditions(self, max_iters, psc=True, pisc=False):
        check(lib.lutldpc_codec_set_exit_conditions(self._h, int(max_iters), int(psc), int(pisc)))

    def set_initial_message_mode(self, mode: int):
        check(lib.lutldpc_codec_set_initial_message_mode(self._h, int(mode)))

    def save(self, path):
        check(lib.lutldpc_codec_save(self._h, str(path).encode()))

    # ---- getters --------------------------------------------------------------------------------
    def graph(self):
        dv, dc, cn = np.zeros(self.nvar, np.int32), np.zeros(self.nchk, np.int32), np.zeros(self.nedges, np.int32)
        check(lib.lutldpc_codec_graph(self._h, _p(dv, C.c_int32), _p(dc, C.c_int32), _p(cn, C.c_int32)))
        return dv, dc, cn

    def _txt(self, fn) -> str:
        need = fn(self._h, None, 0)
        buf = C.create_string_buffer(int(need))
        fn(self._h, buf, need)
        return buf.value.decode()

    @property
    def var_trees_txt(self) -> str:
        return self._txt(lib.lutldpc_codec_var_trees_txt)

    @property
    def chk_trees_txt(self) -> str:
        return self._txt(lib.lutldpc_codec_chk_trees_txt)

    def qb(self, which: int) -> np.ndarray:
        n = lib.lutldpc_codec_qb(self._h, which, None, 0)
        out = np.zeros(n, np.float64)
        lib.lutldpc_codec_qb(self._h, which, _p(out, C.c_double), n)
        return out

    @property
    def qb_cha(self):
        return self.qb(0)

    @property
    def qb_msg(self):
        return self.qb(1)

    @property
    def cha2msg_map(self) -> np.ndarray:
        n = lib.lutldpc_codec_cha2msg_map(self._h, None, 0)
        out = np.zeros(n, np.int32)
        lib.lutldpc_codec_cha2msg_map(self._h, _p(out, C.c_int32), n)
        return out

    @property
    def rate(self) -> float:
        return lib.lutldpc_codec_rate(self._h)

    def decoder(self) -> Decoder:
        """The HIP decoder behind this codec (borrowed handle, for device-pointer decode and profiling)."""
        h = lib.lutldpc_codec_decoder(self._h)
        if not h:
            check(-5)
        d = Decoder.__new__(Decoder)
        d._h = C.c_void_p(h)
        d.nvar, d.nchk, d.max_iters, d.device = self.nvar, self.nchk, self.max_iters, 0
        d._owned = False            # the codec owns (and destroys) the handle
        d._owner = self
        return d

    # ---- decode -----------------------------------------------------------------------------------
    def decode_llr_batch(self, llr):
        llr = np.ascontiguousarray(llr, np.float64)
        B, N = llr.shape
        bits, iters = np.empty((B, N), np.uint8), np.empty(B, np.int32)
        check(lib.lutldpc_codec_decode_llr_batch(self._h, _p(llr, C.c_double), B, _p(bits, C.c_uint8), _p(iters, C.c_int32)))
        return bits, iters

    def lut_decode_dump(self, cha, msg0, level=2):
        """lut_decode with output_verbosity = level (2 or 3): (bits, iters, text of the message dumps as the reference prints them)."""
        cha = np.ascontiguousarray(cha, np.uint8); msg0 = np.ascontiguousarray(msg0, np.uint8)
        B = cha.shape[0]
        bits = np.empty_like(cha); iters = np.empty(B, np.int32)
        n = lib.lutldpc_codec_lut_decode_dump(self._h, _p(cha, C.c_uint8), _p(msg0, C.c_uint8), B, int(level), _p(bits, C.c_uint8), _p(iters, C.c_int32), None, 0)
        if n < 0:
            check(int(n))
        buf = C.create_string_buffer(n)
        n2 = lib.lutldpc_codec_lut_decode_dump(self._h, _p(cha, C.c_uint8), _p(msg0, C.c_uint8), B, int(level), _p(bits, C.c_uint8), _p(iters, C.c_int32), buf, n)
        if n2 < 0:
            check(int(n2))
        return bits, iters, buf.value.decode()

    def lut_decode_batch(self, cha, msg0):
        cha = np.ascontiguousarray(cha, np.uint8)
        msg0 = np.ascontiguousarray(msg0, np.uint8)
        B, N = cha.shape
        bits, iters = np.empty((B, N), np.uint8), np.empty(B, np.int32)
        check(lib.lutldpc_codec_lut_decode_batch(self._h, _p(cha, C.c_uint8), _p(msg0, C.c_uint8), B, _p(bits, C.c_uint8), _p(iters, C.c_int32)))
        return bits, iters

    # ---- Monte-Carlo front end ------------------------------------------------------------------------
    def sim_batch(self, snr_db, seed, stream, frame0, B, zero_codeword=True) -> np.ndarray:
        """{iters, frame error, data bit errors, uncoded errors} of frames frame0..frame0+B-1 (device sampler + decode)."""
        stats = np.empty((B, 4), np.int32)
        check(lib.lutldpc_codec_sim_batch(self._h, float(snr_db), int(seed), int(stream), int(frame0), int(B), int(zero_codeword), _p(stats, C.c_int32)))
        return stats

    def sample_labels(self, snr_db, seed, stream, frame0, B, zero_codeword=True):
        cha, msg = np.empty((B, self.nvar), np.uint8), np.empty((B, self.nvar), np.uint8)
        cw = np.empty((B, self.nvar), np.uint8)
        check(lib.lutldpc_codec_sample_labels(self._h, float(snr_db), int(seed), int(stream), int(frame0), int(B), int(zero_codeword),
                                              _p(cha, C.c_uint8), _p(msg, C.c_uint8), _p(cw, C.c_uint8)))
        return cha, msg, cw

    def channel_cells(self, snr_db):
        thr = np.zeros(72, np.uint64)
        arrs = [np.zeros(72, np.uint8) for _ in range(5)]
        n = lib.lutldpc_codec_channel_cells(self._h, float(snr_db), _p(thr, C.c_uint64), *[_p(a, C.c_uint8) for a in arrs])
        check(min(n, 0))
        return {"thr": thr[:n - 1], "cha": arrs[0][:n], "msg": arrs[1][:n], "neg": arrs[2][:n], "cha_m": arrs[3][:n], "msg_m": arrs[4][:n]}

    def encode(self, info):
        info = np.ascontiguousarray(info, np.uint8)
        assert info.shape == (self.ninfo,)
        cw = np.empty(self.nvar, np.uint8)
        check(lib.lutldpc_codec_encode(self._h, _p(info, C.c_uint8), _p(cw, C.c_uint8)))
        return cw


def de_threshold(dl, lam, dr, rho, qbits_cha=4, qbits_msg=4, maxiter_de=2000, min_lut=True, tree_mode="auto_bin_balanced",
                 strategy="joint_root", thr_min=1e-7, thr_prec=1e-5, pe_max=1e-10, maxiter_bisec=50, max_ni_de_iters=1,
                 llr_max=25.0, nq_fine=5000):
    dl = np.ascontiguousarray(dl, np.int32)
    dr = np.ascontiguousarray(dr, np.int32)
    lam = np.ascontiguousarray(lam, np.float64)
    rho = np.ascontiguousarray(rho, np.float64)
    thr = C.c_double()
    it = lib.lutldpc_de_threshold(_p(dl, C.c_int32), _p(lam, C.c_double), len(dl), _p(dr, C.c_int32), _p(rho, C.c_double), len(dr),
                                  qbits_cha, qbits_msg, maxiter_de, int(min_lut), tree_mode.encode(), strategy.encode(), thr_min,
                                  thr_prec, pe_max, maxiter_bisec, max_ni_de_iters, llr_max, nq_fine, C.byref(thr))
    if it < -1:
        check(it)
    return thr.value, it
