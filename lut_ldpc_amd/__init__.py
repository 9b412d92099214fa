"""lut_ldpc_amd -- MI355X-native LUT-LDPC decode path (drop-in for the hot path of mmeidlinger/lut_ldpc).

The product is the shared library `lut_ldpc_amd/lib/liblut_ldpc_amd.so` (hand-written HIP kernels
for gfx950 behind the C-ABI of `include/lut_ldpc_hip.h` plus the C++ host mirror of
LDPC_Code_LUT / LDPC_BER_Sim_LUT) and the `ber_sim` CLI next to it.  This package only binds it.
"""
from ._capi import LutLdpcError, device_count, last_error, LIB_PATH  # noqa: F401
from .decoder import Decoder  # noqa: F401
from .codec import Codec, de_threshold  # noqa: F401
from .bp import BPDecoder  # noqa: F401

__all__ = ["Decoder", "Codec", "BPDecoder", "de_threshold", "LutLdpcError", "device_count", "last_error", "LIB_PATH"]
