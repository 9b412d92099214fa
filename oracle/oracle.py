"""ctypes loader for the CPU oracle (oracle/*.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product package ``lut_ldpc_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liblut_ldpc_oracle.so"
_OVERRIDE = os.environ.get("LUTLDPC_ORACLE_LIB")      # a sanitizer build of the same sources (tests/fakehip/Makefile)


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (seconds)."""
    if force or not _LIB_PATH.exists() or any(
        p.stat().st_mtime > _LIB_PATH.stat().st_mtime for p in list(_HERE.glob("*.c")) + list(_HERE.glob("*.h"))
    ):
        subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _OVERRIDE:
            build()
        L = C.CDLL(_OVERRIDE or str(_LIB_PATH))
        vp, ip, dp, cp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_char_p
        u8p = C.POINTER(C.c_uint8)
        sig = {
            "or_free": (None, [vp]),
            "or_api_readme_tree": (vp, [cp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
            "or_code_load_alist": (vp, [cp]),
            "or_code_free": (None, [vp]),
            "or_code_gf2_rank": (C.c_int, [vp]),
            "or_api_code_dims": (None, [vp, ip, ip, ip]),
            "or_api_code_graph": (None, [vp, ip, ip, ip]),
            "or_api_code_rows": (None, [vp, ip]),
            "or_codec_new": (vp, [vp, C.c_int]),
            "or_codec_free": (None, [vp]),
            "or_codec_design_luts": (C.c_double, [vp, cp, C.c_int, C.c_double, C.c_int, u8p, C.c_int, ip, C.c_int]),
            "or_codec_set_trees_txt": (C.c_int, [vp, cp, cp, C.c_int, u8p, C.c_int, ip, C.c_int]),
            "or_codec_set_exit_conditions": (None, [vp, C.c_int, C.c_int, C.c_int]),
            "or_codec_lut_decode_batch_u8": (None, [vp, u8p, u8p, C.c_int, u8p, C.POINTER(C.c_int32)]),
            "or_codec_lut_decode_dump": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), u8p, C.c_int, C.c_char_p]),
            "or_codec_decode_llr": (C.c_int, [vp, dp, u8p, ip, ip]),
            "or_sim_awgn_llr": (None, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_double, u8p, dp, ip]),
            "or_bp_new": (vp, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
            "or_bp_free": (None, [vp]),
            "or_bp_set_exit_conditions": (None, [vp, C.c_int, C.c_int, C.c_int]),
            "or_bp_table": (C.c_int, [vp, ip]),
            "or_bp_decode_qllr_batch": (None, [vp, ip, C.c_int, u8p, C.POINTER(C.c_int32), ip]),
            "or_bp_decode_llr_batch": (None, [vp, dp, C.c_int, u8p, C.POINTER(C.c_int32), ip]),
            "or_flat_new": (vp, [vp]),
            "or_flat_free": (None, [vp]),
            "or_flat_decode_batch_u8": (None, [vp, u8p, u8p, C.c_int, u8p, C.POINTER(C.c_int32), C.c_int]),
            "or_codec_syndrome_ok": (C.c_int, [vp, u8p]),
            "or_api_codec_ninfo": (C.c_int, [vp]),
            "or_api_codec_rank": (C.c_int, [vp]),
            "or_api_codec_set_rank": (None, [vp, C.c_int]),
            "or_api_codec_set_initial_message_mode": (None, [vp, C.c_int]),
            "or_api_codec_var_tree_txt": (vp, [vp]),
            "or_api_codec_chk_tree_txt": (vp, [vp]),
            "or_api_codec_qb": (C.c_int, [vp, C.c_int, dp]),
            "or_api_codec_cha2msg_map": (C.c_int, [vp, ip]),
            "or_api_codec_tree_eval": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, ip, C.c_int, ip]),
            "or_api_codec_tree_info": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, ip, ip]),
            "or_api_codec_n_sets": (C.c_int, [vp, C.c_int]),
            "or_sim_channel_cells": (C.c_int, [vp, C.c_double, C.c_double, C.POINTER(C.c_uint64), u8p, u8p, u8p, u8p, u8p]),
            "or_sim_sample_labels": (None, [vp, C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, u8p, u8p, u8p, ip]),
            "or_sim_snr_point": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint32, C.c_int64, C.c_int, C.c_double,
                                           C.c_double, u8p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
            "or_sim_info_bits": (None, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, u8p]),
            "or_code_from_graph": (vp, [C.c_int, C.c_int, ip, ip, ip]),
            "or_api_quant_nonlin_vec": (None, [dp, C.c_int, dp, C.c_int, u8p]),
            "or_api_de_threshold": (C.c_int, [ip, dp, C.c_int, ip, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, cp, cp,
                                              C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, dp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _take_str(ptr) -> str:
    if not ptr:
        raise RuntimeError("oracle returned NULL")
    s = C.string_at(ptr).decode()
    lib().or_free(ptr)
    return s


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def readme_tree(tmpl="riim/im/m///iim/m//im/m////c//", m1=3.0, m2=2.0, nq_in=16, nq_out=16, nq_cha=16) -> str:
    """Design the tree of the reference's trees/README.md:24-43 and serialise it."""
    return _take_str(lib().or_api_readme_tree(tmpl.encode(), m1, m2, nq_in, nq_out, nq_cha))


class Code:
    """Parity-check matrix loaded from an alist file (oracle side), or rebuilt from graph arrays."""

    def __init__(self, alist_path=None, graph=None):
        if graph is not None:
            nvar, nchk, dv, dc, cn = graph
            dv, dc, cn = (np.ascontiguousarray(a, np.int32) for a in (dv, dc, cn))
            self._h = lib().or_code_from_graph(int(nvar), int(nchk), _ip(dv), _ip(dc), _ip(cn))
        else:
            self._h = lib().or_code_load_alist(str(alist_path).encode())
        if not self._h:
            raise FileNotFoundError(f"cannot load alist {alist_path}")
        n, m, e = C.c_int(), C.c_int(), C.c_int()
        lib().or_api_code_dims(self._h, n, m, e)
        self.nvar, self.nchk, self.nedges = n.value, m.value, e.value
        self.dv = np.zeros(self.nvar, np.int32)
        self.dc = np.zeros(self.nchk, np.int32)
        self.cn_msg_idx = np.zeros(self.nedges, np.int32)
        lib().or_api_code_graph(self._h, _ip(self.dv), _ip(self.dc), _ip(self.cn_msg_idx))
        self.row_idx = np.zeros(self.nedges, np.int32)
        lib().or_api_code_rows(self._h, _ip(self.row_idx))

    def rank(self) -> int:
        return lib().or_code_gf2_rank(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().or_code_free(self._h)
            self._h = None


class Codec:
    """Oracle counterpart of the reference's LDPC_Code_LUT."""

    def __init__(self, code: Code, skip_rank: bool = True):
        self.code = code
        self._h = lib().or_codec_new(code._h, int(skip_rank))
        self.max_iters = 0
        self.nq_cha = 0
        self.nq_msg = None
        self.reuse_vec = None
        self.min_lut = True

    def __del__(self):
        if getattr(self, "_h", None):
            lib().or_codec_free(self._h)
            self._h = None

    def design_luts(self, tree_method="auto_bin_balanced", min_lut=True, sigma2=0.88 ** 2, max_iters=50,
                    reuse_vec=None, nq_cha=16, nq_msg=None, allow_deg1=False) -> float:
        reuse = np.zeros(max_iters, np.uint8) if reuse_vec is None else np.asarray(reuse_vec, np.uint8)
        nq = np.full(max_iters, 16, np.int32) if nq_msg is None else np.asarray(nq_msg, np.int32)
        assert len(reuse) == max_iters and len(nq) == max_iters
        sig = lib().or_codec_design_luts(self._h, tree_method.encode(), int(min_lut), float(sigma2), max_iters,
                                         _u8p(reuse), nq_cha, _ip(nq), int(allow_deg1))
        if sig < 0:
            raise RuntimeError("oracle design_luts failed (reference would abort, e.g. degree-1 VN with auto trees)")
        self.max_iters, self.nq_cha, self.nq_msg, self.reuse_vec, self.min_lut = max_iters, nq_cha, nq, reuse, bool(min_lut)
        return sig

    def set_trees_txt(self, var_txt, chk_txt, max_iters, reuse_vec, nq_cha, nq_msg, min_lut):
        reuse = np.asarray(reuse_vec, np.uint8)
        nq = np.asarray(nq_msg, np.int32)
        rc = lib().or_codec_set_trees_txt(self._h, var_txt.encode(), (chk_txt or "").encode(), max_iters, _u8p(reuse),
                                          nq_cha, _ip(nq), int(min_lut))
        if rc != 0:
            raise RuntimeError("oracle set_trees failed")
        self.max_iters, self.nq_cha, self.nq_msg, self.reuse_vec, self.min_lut = max_iters, nq_cha, nq, reuse, bool(min_lut)

    def set_exit_conditions(self, max_iters, psc=True, pisc=False):
        lib().or_codec_set_exit_conditions(self._h, max_iters, int(psc), int(pisc))

    def set_initial_message_mode(self, mode: int):
        lib().or_api_codec_set_initial_message_mode(self._h, mode)

    @property
    def var_tree_txt(self) -> str:
        return _take_str(lib().or_api_codec_var_tree_txt(self._h))

    @property
    def chk_tree_txt(self) -> str:
        return _take_str(lib().or_api_codec_chk_tree_txt(self._h))

    def qb(self, which: int) -> np.ndarray:
        n = lib().or_api_codec_qb(self._h, which, None)
        out = np.zeros(n, np.float64)
        lib().or_api_codec_qb(self._h, which, _dp(out))
        return out

    @property
    def qb_cha(self):
        return self.qb(0)

    @property
    def qb_msg(self):
        return self.qb(1)

    @property
    def cha2msg_map(self) -> np.ndarray:
        n = lib().or_api_codec_cha2msg_map(self._h, None)
        out = np.zeros(n, np.int32)
        lib().or_api_codec_cha2msg_map(self._h, _ip(out))
        return out

    @property
    def rank(self) -> int:
        return lib().or_api_codec_rank(self._h)

    def set_rank(self, r: int):
        lib().or_api_codec_set_rank(self._h, r)

    def lut_decode_batch(self, cha: np.ndarray, msg0: np.ndarray):
        """cha, msg0: uint8 [B, nvar] -> (bits uint8 [B, nvar], iters int32 [B])"""
        cha = np.ascontiguousarray(cha, np.uint8)
        msg0 = np.ascontiguousarray(msg0, np.uint8)
        B, N = cha.shape
        assert N == self.code.nvar and msg0.shape == cha.shape
        out = np.zeros((B, N), np.uint8)
        iters = np.zeros(B, np.int32)
        lib().or_codec_lut_decode_batch_u8(self._h, _u8p(cha), _u8p(msg0), B, _u8p(out),
                                           iters.ctypes.data_as(C.POINTER(C.c_int32)))
        return out, iters

    def lut_decode_dump(self, cha: np.ndarray, msg0: np.ndarray, level: int):
        """Frame by frame with output_verbosity = level (2 or 3): (bits, iters, the text the reference streams to std::cout,
        src/LDPC_Code_LUT.cpp:292-298,311-317,331-337)."""
        import os, tempfile
        cha = np.ascontiguousarray(cha, np.int32)
        msg0 = np.ascontiguousarray(msg0, np.int32)
        B, N = cha.shape
        out = np.zeros((B, N), np.uint8)
        iters = np.zeros(B, np.int32)
        fd, path = tempfile.mkstemp(suffix=".txt")
        os.close(fd)
        try:
            for f in range(B):
                iters[f] = lib().or_codec_lut_decode_dump(self._h, cha[f].ctypes.data_as(C.POINTER(C.c_int)), msg0[f].ctypes.data_as(C.POINTER(C.c_int)),
                                                         _u8p(out[f]), int(level), path.encode())
            text = open(path).read()
        finally:
            os.unlink(path)
        return out, iters, text

    def lut_decode_batch_flat(self, cha: np.ndarray, msg0: np.ndarray, threads: int = 0):
        """Flat-table mode (or_flat.c): same results, trees flattened to arrays, one frame per thread (0 = all cores)."""
        cha = np.ascontiguousarray(cha, np.uint8)
        msg0 = np.ascontiguousarray(msg0, np.uint8)
        B, N = cha.shape
        assert N == self.code.nvar and msg0.shape == cha.shape
        out = np.zeros((B, N), np.uint8)
        iters = np.zeros(B, np.int32)
        F = lib().or_flat_new(self._h)              # (re-flattened per call: the trees may have been replaced)
        try:
            lib().or_flat_decode_batch_u8(F, _u8p(cha), _u8p(msg0), B, _u8p(out), iters.ctypes.data_as(C.POINTER(C.c_int32)),
                                          int(threads) or (os.cpu_count() or 1))
        finally:
            lib().or_flat_free(F)
        return out, iters

    def tree_eval(self, kind: int, tree_set: int, cls: int, inputs, n_out: int) -> np.ndarray:
        a = np.ascontiguousarray(inputs, np.int32)
        out = np.zeros(n_out, np.int32)
        if lib().or_api_codec_tree_eval(self._h, kind, tree_set, cls, _ip(a), len(a), _ip(out)) != 0:
            raise IndexError("no such tree")
        return out

    def tree_info(self, kind: int, tree_set: int, cls: int):
        t, n = C.c_int(), C.c_int()
        if lib().or_api_codec_tree_info(self._h, kind, tree_set, cls, t, n) != 0:
            raise IndexError("no such tree")
        return t.value, n.value

    def n_sets(self, chk=False) -> int:
        return lib().or_api_codec_n_sets(self._h, int(chk))

    # ---- Monte-Carlo front end (oracle/or_sim.c) ---------------------------------------------------
    def channel_cells(self, snr_db: float, rate: float):
        thr = np.zeros(80, np.uint64)
        arrs = [np.zeros(80, np.uint8) for _ in range(5)]
        n = lib().or_sim_channel_cells(self._h, snr_db, rate, thr.ctypes.data_as(C.POINTER(C.c_uint64)), *[_u8p(a) for a in arrs])
        return {"thr": thr[:n - 1], "cha": arrs[0][:n], "msg": arrs[1][:n], "neg": arrs[2][:n], "cha_m": arrs[3][:n], "msg_m": arrs[4][:n]}

    def sample_labels(self, snr_db, rate, seed, stream, frame0, B, codewords=None):
        N = self.code.nvar
        cha, msg = np.zeros((B, N), np.uint8), np.zeros((B, N), np.uint8)
        unc = np.zeros(B, np.int32)
        cw = None if codewords is None else np.ascontiguousarray(codewords, np.uint8)
        lib().or_sim_sample_labels(self._h, snr_db, rate, seed, stream, frame0, B, _u8p(cw) if cw is not None else None, _u8p(cha), _u8p(msg), _ip(unc))
        return cha, msg, unc

    def sim_snr_point(self, snr_db, rate, K, seed, stream, nframes, nfers=20, ber_min=1e-7, fer_min=1e-5, codewords=None):
        counters = np.zeros(5, np.int64)
        per = np.zeros((nframes, 4), np.int32)
        cw = None if codewords is None else np.ascontiguousarray(codewords, np.uint8)
        stop = lib().or_sim_snr_point(self._h, snr_db, rate, K, seed, stream, nframes, nfers, ber_min, fer_min,
                                      _u8p(cw) if cw is not None else None, counters.ctypes.data_as(C.POINTER(C.c_int64)),
                                      per.ctypes.data_as(C.POINTER(C.c_int32)))
        return counters, per[:counters[0]], bool(stop)

    def syndrome_ok(self, bits: np.ndarray) -> bool:
        b = np.ascontiguousarray(bits, np.uint8)
        return bool(lib().or_codec_syndrome_ok(self._h, _u8p(b)))


def info_bits(seed, stream, frame, K) -> np.ndarray:
    out = np.zeros(K, np.uint8)
    lib().or_sim_info_bits(seed, stream, frame, K, _u8p(out))
    return out


def quant_nonlin(x: np.ndarray, bounds: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float64)
    b = np.ascontiguousarray(bounds, np.float64)
    out = np.zeros(x.shape, np.uint8)
    lib().or_api_quant_nonlin_vec(_dp(x), x.size, _dp(b), b.size, _u8p(out))
    return out


def de_threshold(dl, lam, dr, rho, qbits_cha=4, qbits_msg=4, maxiter_de=2000, min_lut=True,
                 tree_mode="auto_bin_balanced", strategy="joint_root", thr_min=1e-7, thr_prec=1e-5, pe_max=1e-10,
                 maxiter_bisec=50, max_ni_de_iters=1, llr_max=25.0, nq_fine=5000):
    dl = np.asarray(dl, np.int32); dr = np.asarray(dr, np.int32)
    lam = np.asarray(lam, np.float64); rho = np.asarray(rho, np.float64)
    thr = C.c_double()
    it = lib().or_api_de_threshold(_ip(dl), _dp(lam), len(dl), _ip(dr), _dp(rho), len(dr), qbits_cha, qbits_msg,
                                   maxiter_de, int(min_lut), tree_mode.encode(), strategy.encode(), thr_min, thr_prec,
                                   pe_max, maxiter_bisec, max_ni_de_iters, llr_max, nq_fine, C.byref(thr))
    return thr.value, it


class BP:
    """[BP] comparison decoder of the oracle (or_bp.c): the specification of include/lut_ldpc_bp.h on the CPU.  PARITY
    UNPINNED against the reference's forked IT++ (absent)."""

    def __init__(self, code: "Code", d1=12, d2=300, d3=7, d4=28):
        self.code = code
        self._h = lib().or_bp_new(code._h, d1, d2, d3, d4)
        self.d2 = d2

    def __del__(self):
        if getattr(self, "_h", None):
            lib().or_bp_free(self._h)
            self._h = None

    def set_exit_conditions(self, max_iters, psc=True, pisc=False):
        lib().or_bp_set_exit_conditions(self._h, int(max_iters), int(psc), int(pisc))

    def table(self) -> np.ndarray:
        out = np.zeros(max(self.d2, 1), np.int32)
        n = lib().or_bp_table(self._h, _ip(out))
        return out[:n]

    def _decode(self, fn, a, ptr):
        B, N = a.shape
        bits, iters, q = np.zeros((B, N), np.uint8), np.zeros(B, np.int32), np.zeros((B, N), np.int32)
        fn(self._h, ptr, B, _u8p(bits), iters.ctypes.data_as(C.POINTER(C.c_int32)), _ip(q))
        return bits, iters, q

    def decode_llr_batch(self, llr):
        a = np.ascontiguousarray(llr, np.float64)
        return self._decode(lib().or_bp_decode_llr_batch, a, a.ctypes.data_as(C.POINTER(C.c_double)))

    def decode_qllr_batch(self, qllr):
        a = np.ascontiguousarray(qllr, np.int32)
        return self._decode(lib().or_bp_decode_qllr_batch, a, _ip(a))


def awgn_llr(seed, stream, frame0, B, N, N0, codewords=None):
    """BPSK / AWGN LLRs of frames frame0..frame0+B-1 ([BP] front end, or_sim.c): (llr [B, N] float64, uncoded errors [B])."""
    llr, unc = np.zeros((B, N), np.float64), np.zeros(B, np.int32)
    cw = None if codewords is None else np.ascontiguousarray(codewords, np.uint8)
    lib().or_sim_awgn_llr(int(seed), int(stream), int(frame0), int(B), int(N), float(N0), _u8p(cw) if cw is not None else None,
                          llr.ctypes.data_as(C.POINTER(C.c_double)), _ip(unc))
    return llr, unc
