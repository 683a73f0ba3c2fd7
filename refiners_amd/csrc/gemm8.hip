// Instantiations of the 8-wave / eight-phase main loop (gemm8_kernel.cuh): plain GEMM and implicit-GEMM convolution, bf16 and f32 (parity mode);
// entered through mi355x_gemm (gemm.hip) for tile configuration 7.
#include "gemm8_kernel.cuh"

namespace mi355x {
int launch_gemm8_f32(const GemmP& p, hipStream_t stream, bool streamk, int mt) { return launch_gemm8<float, false>(p, stream, streamk, mt); }
int launch_gemm8_bf16(const GemmP& p, hipStream_t stream, bool streamk, int mt) { return launch_gemm8<bf16_t, false>(p, stream, streamk, mt); }
int launch_conv8_f32(const GemmP& p, hipStream_t stream, bool streamk, int mt) { return launch_gemm8<float, true>(p, stream, streamk, mt); }
int launch_conv8_bf16(const GemmP& p, hipStream_t stream, bool streamk, int mt) { return launch_gemm8<bf16_t, true>(p, stream, streamk, mt); }
}  // namespace mi355x
