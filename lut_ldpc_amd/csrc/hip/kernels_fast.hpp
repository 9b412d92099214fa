// kernels_fast.hpp -- specialised gfx950 kernels for the shapes ber_sim actually produces:
// min-sum check nodes of small degree and balanced binary variable/decision trees.
// A launcher returns 1 when it handled the whole pass, 0 when the pass must go to the generic
// kernels, <0 on a launch error.
#pragma once
#include "kernels_common.hpp"
#include "lut_program.hpp"

namespace lutldpc {

template <int KIND>
inline int launch_fast_tree_pass(hipStream_t, const PassParams &, bool, uint8_t *, const uint8_t *, uint8_t *, const uint32_t *,
                                 uint32_t *, const Op *, const uint8_t *, const int32_t *, const int32_t *) {
    return 0;
}

inline int launch_fast_cn_minsum(hipStream_t, const PassParams &, uint8_t *, const uint32_t *, uint32_t *, const int32_t *,
                                 const int32_t *, const int32_t *) {
    return 0;
}

inline const char *fast_vn_kernel_name(int, int, const Program *) { return "tree_pass_kernel<VAR>"; }
inline const char *fast_cn_kernel_name(int, int) { return "cn_minsum_generic_kernel"; }

}  // namespace lutldpc
