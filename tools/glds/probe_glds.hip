// Microbenchmark: how fast can one CU stream global memory into LDS with global_load_lds_dwordx4, as a function of the
// ADDRESS PATTERN of a wave instruction?  (The GEMM's loader reads 8 rows x 128 B per instruction at the operand's row stride;
// a tile-packed operand would be 1 KB contiguous.)  No MFMA, no LDS reads: issue one stage, wait, barrier, next stage.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// mode 0: row-strided (8 rows x 128 B per wave instruction, rows `ld` bytes apart, K block kb advances by 128 B)
// mode 1: contiguous (each stage is one 32 KB contiguous block)
// mode 2: row-strided with 256-B rows (4 rows x 256 B per instruction: BK = 128)
template <int STAGES>
__global__ __launch_bounds__(256) void glds_probe(const char* __restrict__ base, int64_t bytes, int64_t ld, int mode, int iters, int tiles_per_wg, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int STAGE = 32768;  // 256 rows x 128 B (a 128x128 GEMM tile's two operands for one 64-wide K block)
    int issued = 0;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int64_t tile = (int64_t)blockIdx.x * tiles_per_wg + t;
        for (int kb = 0; kb < iters; ++kb) {
            char* dst = smem + (issued % STAGES) * STAGE;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int q = it * 256 + tid;
                int64_t off;
                if (mode == 0) {
                    const int row = q >> 3, ch = q & 7;
                    off = ((tile * 256 + row) * ld + (int64_t)kb * 128 + ch * 16);
                } else if (mode == 1) {
                    off = ((tile * iters + kb) * STAGE + (int64_t)q * 16);
                } else {
                    const int row = q >> 4, ch = q & 15;
                    off = ((tile * 128 + row) * ld + (int64_t)kb * 256 + ch * 16);
                }
                off %= bytes;
                __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(dst + (it * 256 + wid * 64) * 16), 16, 0, 0);
            }
            ++issued;
            if (STAGES == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // one stage may stay in flight
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && smem[17] == 123 && bytes < 0) *sink = 1;
}

extern "C" int glds_probe_run(const void* base, int64_t bytes, int64_t ld, int mode, int iters, int tiles_per_wg, int grid, int stages, int lds_bytes, int* sink,
                              void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stages == 2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(glds_probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(glds_probe<2>, dim3(grid), dim3(256), lds_bytes, st, static_cast<const char*>(base), bytes, ld, mode, iters, tiles_per_wg, sink);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(glds_probe<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(glds_probe<3>, dim3(grid), dim3(256), lds_bytes, st, static_cast<const char*>(base), bytes, ld, mode, iters, tiles_per_wg, sink);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
