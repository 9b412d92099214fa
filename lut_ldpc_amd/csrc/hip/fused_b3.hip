// fused_b3.hip -- explicit instantiation of the fused pass kernel, degree bucket 3 (see kernels_fast.hpp)
#include "kernels_fast.hpp"
namespace lutldpc {
template void launch_fused<1, 3> LUTLDPC_FUSED_SIG;
template void launch_fused<2, 3> LUTLDPC_FUSED_SIG;
template hipError_t preload_fused<2, 3>();
}
