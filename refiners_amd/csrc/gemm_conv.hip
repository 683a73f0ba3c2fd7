// Implicit-GEMM convolution instantiations of gemm_kernel.cuh (CONV = true); entered through mi355x_gemm (gemm.hip).
#include "gemm_kernel.cuh"

namespace mi355x {
int launch_conv_f32(const GemmP& p, hipStream_t stream) { return launch_tile<float, true>(p, stream); }
int launch_conv_bf16(const GemmP& p, hipStream_t stream) { return launch_tile<bf16_t, true>(p, stream); }
}  // namespace mi355x
