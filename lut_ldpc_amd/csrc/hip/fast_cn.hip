// fast_cn.hip -- explicit instantiation of one group of specialised-kernel launchers (see kernels_fast.hpp)
#include "kernels_fast.hpp"
namespace lutldpc {
template bool launch_cn_fast<1>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint32_t *, uint32_t *, const int32_t *);
template bool launch_cn_fast<2>(hipStream_t, const FastParams &, const FastParams *, uint8_t *, const uint32_t *, uint32_t *, const int32_t *);
template hipError_t preload_cn_fast<2>();
}
