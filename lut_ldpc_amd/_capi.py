"""ctypes binding of the C-ABI declared in include/lut_ldpc_hip.h (liblut_ldpc_amd.so).

There is no CPU fallback: if the shared library is missing this module raises at import, and
every decode call fails with the library's error text when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "liblut_ldpc_amd.so"
if os.environ.get("LUTLDPC_LIB"):          # A/B runs of two builds of the same library (tools/)
    LIB_PATH = Path(os.environ["LUTLDPC_LIB"])

OK, ERR_ARG, ERR_PARSE, ERR_UNSUPPORTED, ERR_HIP, ERR_STATE = 0, -1, -2, -3, -4, -5
K_CN_PASS, K_VN_PASS, K_DECISION, K_SYNDROME, K_LAYOUT, K_FRONTEND, K_FUSED_PASS, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7
KIND_NAMES = ["cn_pass", "vn_pass", "decision", "syndrome", "layout", "frontend", "fused_pass", "resident"]


class LutLdpcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lut_ldpc_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
        )
    return C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _ip, _u8p, _dp, _cp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.c_char_p
_SIGNATURES = {
    "lutldpc_last_error": (_cp, []),
    "lutldpc_version": (_cp, []),
    "lutldpc_device_count": (C.c_int, []),
    "lutldpc_decoder_create": (C.c_int, [C.c_int, C.c_int, _ip, _ip, _ip, C.c_int, _ip, _u8p, C.c_int, C.c_int, _cp, _cp, C.c_int,
                                         C.POINTER(_vp)]),
    "lutldpc_decoder_destroy": (C.c_int, [_vp]),
    "lutldpc_decoder_set_exit_conditions": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "lutldpc_decoder_decode_batch": (C.c_int, [_vp, _u8p, _u8p, C.c_int, _u8p, _ip]),
    "lutldpc_decoder_decode_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int]),
    "lutldpc_decoder_decode_llr_batch": (C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, _ip, _u8p, _ip]),
    "lutldpc_decoder_stream": (_vp, [_vp]),
    "lutldpc_decoder_set_profiling": (C.c_int, [_vp, C.c_int]),
    "lutldpc_decoder_get_profile": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_int64)]),
    "lutldpc_decoder_reset_profile": (C.c_int, [_vp]),
    "lutldpc_decoder_device_bytes": (C.c_int64, [_vp]),
    "lutldpc_decoder_describe": (_cp, [_vp]),
    "lutldpc_selftest_program_eval": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, C.c_int, _ip, C.c_int]),
    "lutldpc_selftest_program_stats": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip]),
    "lutldpc_decoder_decode_batch_trace": (C.c_int, [_vp, _u8p, _u8p, C.c_int, C.c_int, _u8p, _ip, _u8p, C.c_int64, _ip]),
    "lutldpc_selftest_jit_source": (C.c_int64, [_vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int64, C.c_int]),
    "lutldpc_selftest_resident_source": (C.c_int64, [_vp, C.c_int, C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_int32)]),
}
for _name, (_res, _args) in _SIGNATURES.items():
    if hasattr(lib, _name):
        _fn = getattr(lib, _name)
        _fn.restype, _fn.argtypes = _res, _args


def last_error() -> str:
    return lib.lutldpc_last_error().decode()


def check(rc: int) -> None:
    if rc != OK:
        raise LutLdpcError(rc, last_error())


def device_count() -> int:
    return lib.lutldpc_device_count()

# ---- host mirror (include/lut_ldpc_host.h) ---------------------------------------------------
_HOST_SIGNATURES = {
    "lutldpc_codec_create": (C.c_int, [_cp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lutldpc_codec_load": (C.c_int, [_cp, C.c_int, C.POINTER(_vp)]),
    "lutldpc_codec_save": (C.c_int, [_vp, _cp]),
    "lutldpc_codec_destroy": (C.c_int, [_vp]),
    "lutldpc_codec_design_luts": (C.c_int, [_vp, _cp, C.c_int, C.c_double, C.c_int, _u8p, C.c_int, _ip, C.c_int, _dp]),
    "lutldpc_codec_design_from_cache": (C.c_int, [_vp]),
    "lutldpc_codec_set_exit_conditions": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "lutldpc_codec_set_initial_message_mode": (C.c_int, [_vp, C.c_int]),
    "lutldpc_codec_set_output_verbosity": (C.c_int, [_vp, C.c_int]),
    "lutldpc_codec_dims": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "lutldpc_codec_graph": (C.c_int, [_vp, _ip, _ip, _ip]),
    "lutldpc_codec_var_trees_txt": (C.c_int64, [_vp, C.c_char_p, C.c_int64]),
    "lutldpc_codec_chk_trees_txt": (C.c_int64, [_vp, C.c_char_p, C.c_int64]),
    "lutldpc_codec_qb": (C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    "lutldpc_codec_cha2msg_map": (C.c_int, [_vp, _ip, C.c_int]),
    "lutldpc_codec_rate": (C.c_double, [_vp]),
    "lutldpc_codec_decoder": (_vp, [_vp]),
    "lutldpc_codec_decode_llr_batch": (C.c_int, [_vp, _dp, C.c_int, _u8p, _ip]),
    "lutldpc_codec_lut_decode_batch": (C.c_int, [_vp, _u8p, _u8p, C.c_int, _u8p, _ip]),
    "lutldpc_codec_lut_decode_dump": (C.c_int64, [_vp, _u8p, _u8p, C.c_int, C.c_int, _u8p, _ip, C.c_char_p, C.c_int64]),
    "lutldpc_codec_encode": (C.c_int, [_vp, _u8p, _u8p]),
    "lutldpc_de_threshold": (C.c_int, [_ip, _dp, C.c_int, _ip, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _cp, _cp, C.c_double,
                                       C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, _dp]),
}
for _name, (_res, _args) in _HOST_SIGNATURES.items():
    if hasattr(lib, _name):
        _fn = getattr(lib, _name)
        _fn.restype, _fn.argtypes = _res, _args

_u64p, _i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
_SIM_SIGNATURES = {
    "lutldpc_codec_sim_batch": (C.c_int, [_vp, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int, _ip]),
    "lutldpc_codec_sample_labels": (C.c_int, [_vp, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int, _u8p, _u8p, _u8p]),
    "lutldpc_codec_channel_cells": (C.c_int, [_vp, C.c_double, _u64p, _u8p, _u8p, _u8p, _u8p, _u8p]),
    "lutldpc_ber_sim_run": (C.c_int, [_cp, _cp, C.c_int, _cp, C.c_int, C.c_int, C.c_int, _dp, _i64p, C.c_int]),
    "lutldpc_ber_sim_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "lutldpc_selftest_write_results_it": (C.c_int, [_cp, _dp, _i64p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "lutldpc_bersim_create": (C.c_int, [_cp, _cp, C.c_int, _cp, C.c_int, C.POINTER(_vp)]),
    "lutldpc_bersim_destroy": (C.c_int, [_vp]),
    "lutldpc_bersim_info": (C.c_int, [_vp, _i64p, _dp, _dp, C.c_int]),
    "lutldpc_bersim_batch": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _ip]),
    "lutldpc_bersim_add_point": (C.c_int, [_vp, C.c_double, _i64p]),
    "lutldpc_bersim_save": (C.c_int, [_vp, C.c_double]),
    "lutldpc_bersim_results_path": (C.c_int64, [_vp, C.c_char_p, C.c_int64]),
}
for _name, (_res, _args) in _SIM_SIGNATURES.items():
    if hasattr(lib, _name):
        _fn = getattr(lib, _name)
        _fn.restype, _fn.argtypes = _res, _args
