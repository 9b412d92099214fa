// fast_vn_dec.hip -- explicit instantiation of one group of specialised-kernel launchers (see kernels_fast.hpp)
#include "kernels_fast.hpp"
namespace lutldpc {
template bool launch_vn_fast<TT_DEC, 1>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *);
template bool launch_vn_fast<TT_DEC, 2>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *);
template hipError_t preload_vn_fast<TT_DEC, 2>();
}
