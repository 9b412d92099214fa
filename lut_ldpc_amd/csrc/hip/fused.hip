// fused.hip -- explicit instantiation of one group of specialised-kernel launchers (see kernels_fast.hpp)
#include "kernels_fast.hpp"
namespace lutldpc {
template void launch_fused<1>(hipStream_t, const FusedParams &, const int32_t *, int, bool, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *);
template void launch_fused<2>(hipStream_t, const FusedParams &, const int32_t *, int, bool, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *, uint32_t *, const uint8_t *, const int32_t *);
}
