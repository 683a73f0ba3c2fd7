// Microbenchmark (round 4): what bounds the global -> LDS stream of a GEMM-shaped loop on gfx950?  Every GEMM class of the step lands on
// "bytes staged into LDS / ~11.5 TB/s" whatever the tile; this probe separates the candidates -- a per-CU throughput cap of the
// LDS-DMA path, bytes in flight x latency, the per-stage barrier, L2 hit vs Infinity-Cache / HBM service -- by streaming with NO compute and no LDS reads:
//   * every wave owns a ring of D + 1 slots of S KB in LDS and issues S global_load_lds_dwordx4 (1 KB each, lane-contiguous) per stage;
//   * counted wait: vmcnt(S * D) leaves D stages in flight; BAR = 1 adds the GEMM loop's one s_barrier per stage;
//   * MODE 1 = the same addresses through plain global_load_dwordx4 into registers (no LDS at all), MODE 2 = registers then ds_write_b128;
//   * the working set is per XCD (block b runs on XCD b % 8): small sets are L2 hits after the first pass, large ones come from the
//     Infinity Cache / HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int S, int D, int BAR, int MODE>
__global__ __launch_bounds__(512) void glds_map(const char* __restrict__ base, int64_t set_bytes, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3;
    const char* region = base + (int64_t)xcd * set_bytes;
    char* ring = smem + wid * (D + 1) * S * 1024;
    int64_t off = ((int64_t)(bx * nw + wid) * iters * S) * 1024 % set_bytes;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 hold[MODE ? S * (D + 1) : 1];
#pragma unroll 1
    for (int it = 0; it < iters; it += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {  // D + 1 stages per trip so that ring slots / register sets are compile-time
            if (it + u < iters) {
                const char* stage = region + off + lane * 16;  // (set_bytes is a multiple of S KB: the wrap is per stage, loads use immediate offsets)
                off += S * 1024;
                if (off >= set_bytes) off -= set_bytes;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const char* src = stage + s * 1024;
                    if (MODE == 0) {
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + (u * S + s) * 1024), 16, 0, 0);
                    } else {
                        if (MODE == 2 && it > 0) *reinterpret_cast<f32x4*>(ring + (u * S + s) * 1024 + lane * 16) = hold[u * S + s];  // the stage loaded D + 1 stages ago
                        else acc += hold[u * S + s];
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hold[u * S + s]) : "v"(src) : "memory");
                    }
                }
                wait_vm<S * D>();
                if (BAR) __builtin_amdgcn_s_barrier();
            }
        }
    }
    wait_vm<0>();
    if (MODE) {
#pragma unroll
        for (int i = 0; i < S * (D + 1); ++i) acc += hold[i];
    }
    __syncthreads();
    if (tid == 0 && (smem[17] == 123 || acc[0] == 1.2345f) && set_bytes < 0) *sink = 1;
}

template <int S, int D, int BAR, int MODE>
static int run(const void* base, int64_t set_bytes, int iters, int grid, int threads, int lds, int* sink, hipStream_t st) {
    auto k = glds_map<S, D, BAR, MODE>;
    (void)hipGetLastError();  // (whatever an earlier call of the host framework left behind)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        fprintf(stderr, "hipFuncSetAttribute(%d): %s\n", lds, hipGetErrorString(e));
        return -3;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, st, static_cast<const char*>(base), set_bytes, iters, sink);
    e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "launch S=%d D=%d bar=%d mode=%d grid=%d threads=%d lds=%d: %s\n", S, D, BAR, MODE, grid, threads, lds, hipGetErrorString(e));
        return -1;
    }
    return 0;
}

#define CASE(S_, D_, B_, M_) \
    if (S == S_ && D == D_ && bar == B_ && mode == M_) return run<S_, D_, B_, M_>(base, set_bytes, iters, grid, threads, lds, sink, st);

extern "C" int glds_map_run(const void* base, int64_t set_bytes, int S, int D, int bar, int mode, int iters, int grid, int threads, int lds, int* sink, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    CASE(4, 1, 0, 0) CASE(4, 1, 1, 0) CASE(4, 2, 0, 0) CASE(4, 2, 1, 0) CASE(4, 3, 0, 0) CASE(4, 3, 1, 0)
    CASE(8, 1, 0, 0) CASE(8, 1, 1, 0) CASE(8, 2, 0, 0) CASE(8, 2, 1, 0) CASE(8, 3, 1, 0)
    CASE(2, 1, 1, 0) CASE(2, 3, 1, 0) CASE(2, 7, 1, 0)
    CASE(4, 1, 0, 1) CASE(4, 2, 0, 1) CASE(8, 1, 0, 1) CASE(4, 1, 1, 2) CASE(4, 2, 1, 2)
    return -2;
}
