// Layout / glue kernels of the UNet step: all HBM-bound, grid-stride, 16-byte vectorised where the layout allows.
#include "common.cuh"
#include "../../include/mi355x_refiners.h"

#include <stdio.h>
#include <string.h>

namespace {

inline int grid_for(int64_t work, int per_block = 256, int cap = 4096) {
    int64_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// NCHW -> NHWC through an LDS transpose tile: 64 pixels x 64 channels per workgroup
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* __restrict__ x, T* __restrict__ out, int C, int HW, int64_t ldo) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, px = p0 + tx;
        tile[cc][tx] = (c < C && px < HW) ? to_f32(x[((int64_t)b * C + c) * HW + px]) : 0.f;
    }
    __syncthreads();
    for (int pp = ty; pp < 64; pp += 4) {
        const int px = p0 + pp, c = c0 + tx;
        if (px < HW && c < C) out[((int64_t)b * HW + px) * ldo + c] = from_f32<T>(tile[tx][pp]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, T* __restrict__ out, int C, int HW, int64_t ldx) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int pp = ty; pp < 64; pp += 4) {
        const int px = p0 + pp, c = c0 + tx;
        tile[pp][tx] = (c < C && px < HW) ? to_f32(x[((int64_t)b * HW + px) * ldx + c]) : 0.f;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, px = p0 + tx;
        if (px < HW && c < C) out[((int64_t)b * C + c) * HW + px] = from_f32<T>(tile[tx][cc]);
    }
}

// im2col for a tiny-channel NCHW image (the UNet's 4 -> 320 input conv): one thread per (pixel, column)
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int C, int H, int W,
                                                         int64_t ldo) {
    const int64_t total = (int64_t)B * H * W * ldo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % ldo);
        const int64_t m = i / ldo;
        float v = 0.f;
        if (col < 9 * C) {
            const int tap = col / C, c = col - tap * C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int b = (int)(m / (H * W));
            const int rem = (int)(m - (int64_t)b * H * W);
            const int oy = rem / W, ox = rem - oy * W;
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = to_f32(x[(((int64_t)b * C + c) * H + iy) * W + ix]);
        }
        out[i] = from_f32<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void concat2_kernel(const T* __restrict__ a, int64_t lda, int V1, const T* __restrict__ b, int64_t ldb,
                                                       int V2, T* __restrict__ out, int64_t ldo, int64_t M) {
    constexpr int EPC = DT<T>::EPC;
    const int NV = V1 + V2;
    const int64_t total = M * NV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / NV;
        const int v = (int)(i - m * NV);
        Vec16<T> t = v < V1 ? load16<T>(a + m * lda + v * EPC) : load16<T>(b + m * ldb + (v - V1) * EPC);
        store16<T>(out + m * ldo + v * EPC, t);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ a, float alpha, const T* __restrict__ b, float beta,
                                                     T* __restrict__ out, int64_t n) {
    constexpr int EPC = DT<T>::EPC;
    const int64_t nv = n / EPC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        Vec16<T> x = load16<T>(a + i * EPC), y = load16<T>(b + i * EPC), o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.set(e, alpha * x.get(e) + beta * y.get(e));
        store16<T>(out + i * EPC, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - nv * EPC)) {
        const int64_t i = nv * EPC + threadIdx.x;
        out[i] = from_f32<T>(alpha * to_f32(a[i]) + beta * to_f32(b[i]));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void silu_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = from_f32<T>(silu_f(to_f32(x[i])));
}

template <typename T>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(T* __restrict__ x, const T* __restrict__ uo, const float* __restrict__ coef,
                                                        int64_t n) {
    const float cfg = coef[0], sa = coef[1], s1a = coef[2], sap = coef[3], s1ap = coef[4];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float u = to_f32(uo[i]), c = to_f32(uo[n + i]);
        const float eps = u + cfg * (c - u);
        const float xv = to_f32(x[i]);
        const float x0 = (xv - s1a * eps) / sa;
        x[i] = from_f32<T>(sap * x0 + s1ap * eps);
    }
}

// Classifier-free guidance + any solver whose update is linear in (x, eps, one kept quantity): Euler, DPM-Solver++ 2M, DDIM.
//   eps = u + cfg (c - u);  d = hx x + he eps;  x' = kx x + ke eps + kd d + kp hist;  hist = d;  model_in (both CFG halves) = s_next x'
// coef = {cfg, hx, he, kx, ke, kd, kp, s_next} in device memory (one row of a per-step table, so a captured graph can be replayed)
template <typename T>
__global__ __launch_bounds__(256) void cfg_linear_kernel(T* __restrict__ x, const T* __restrict__ uo, T* __restrict__ hist, T* __restrict__ model_in,
                                                          const float* __restrict__ coef, int64_t n) {
    const float cfg = coef[0], hx = coef[1], he = coef[2], kx = coef[3], ke = coef[4], kd = coef[5], kp = coef[6], sn = coef[7];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float u = to_f32(uo[i]), c = to_f32(uo[n + i]);
        const float eps = u + cfg * (c - u);
        const float xv = to_f32(x[i]);
        const float d = hx * xv + he * eps;
        const float xn = kx * xv + ke * eps + kd * d + kp * to_f32(hist[i]);
        const T xs = from_f32<T>(xn);
        x[i] = xs;
        hist[i] = from_f32<T>(d);
        if (model_in) {
            const T mi = from_f32<T>(sn * to_f32(xs));  // the next step scales the STORED latents, as solver.scale_model_input does
            model_in[i] = mi;
            model_in[n + i] = mi;
        }
    }
}

// out[(i / group) * ldo + col0 + (i % group) * dim + j]       = cos(x[i] * 10000^(-j / half)),  j < half
// out[(i / group) * ldo + col0 + (i % group) * dim + half + j] = sin(...)                         (float32 arithmetic)
template <typename T>
__global__ __launch_bounds__(256) void sinusoidal_kernel(const float* __restrict__ x, int64_t n, int dim, int group, T* __restrict__ out,
                                                          int64_t ldo, int col0) {
    const int half = dim / 2;
    const int64_t total = n * half;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q / half;
        const int j = (int)(q - i * half);
        float e = -9.210340371976184f * (float)j;  // -ln(10000) * j, then / half: the reference's operation order
        e /= (float)half;
        const float a = x[i] * expf(e);
        T* o = out + (i / group) * ldo + col0 + (i % group) * dim;
        o[j] = from_f32<T>(cosf(a));
        o[half + j] = from_f32<T>(sinf(a));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int C, int H, int W, int P, int64_t ldo) {
    const int GH = H / P, GW = W / P, K = C * P * P;
    const int64_t total = (int64_t)B * GH * GW * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % K);
        const int64_t m = i / K;
        const int kx = col % P, ky = (col / P) % P, c = col / (P * P);
        const int px = (int)(m % GW), py = (int)((m / GW) % GH), b = (int)(m / ((int64_t)GW * GH));
        out[m * ldo + col] = x[(((int64_t)b * C + c) * H + py * P + ky) * W + px * P + kx];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, int64_t ldx, const int* __restrict__ idx, T* __restrict__ out,
                                                           int64_t ldo, int64_t n_rows, int NV) {
    constexpr int EPC = DT<T>::EPC;
    const int64_t total = n_rows * NV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / NV;
        const int v = (int)(i - r * NV);
        const int src = idx[r];
        Vec16<T> t;
        if (src >= 0) t = load16<T>(x + (int64_t)src * ldx + v * EPC);
        else
#pragma unroll
            for (int e = 0; e < EPC; ++e) t.set(e, 0.f);
        store16<T>(out + r * ldo + v * EPC, t);
    }
}

// 1x1 convolution of an NCHW image with very few channels (the VAE decoder's 4 -> 4 post-quantisation conv): one thread per
// pixel, weights [Co][Ci] and bias read through the scalar cache.  Ci, Co <= 8.
template <typename T>
__global__ __launch_bounds__(256) void pointwise_nchw_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b, T* __restrict__ out,
                                                              int B, int Ci, int Co, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bi = i / HW, px = i - bi * HW;
        float xin[8];
        for (int c = 0; c < Ci; ++c) xin[c] = to_f32(x[(bi * Ci + c) * HW + px]);
        for (int o = 0; o < Co; ++o) {
            float acc = b ? to_f32(b[o]) : 0.f;
            for (int c = 0; c < Ci; ++c) acc += to_f32(w[o * Ci + c]) * xin[c];
            out[(bi * Co + o) * HW + px] = from_f32<T>(acc);
        }
    }
}

}  // namespace

// P[m][0..L) = softmax(scale * S[m][0..L)), P[m][L..Lp) = 0: one 256-thread workgroup per row, the row held in registers
// (<= 64 values per thread, i.e. L <= 16384) or re-read from L2 in three passes beyond that.  S is float32 (mi355x_gemm with
// out_f32): scores of a 512-wide head reach tens, which bf16 storage would quantise to steps of 0.06-0.12 before the exponential.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds, T* __restrict__ out, int64_t ldo, int L, int Lp, float c) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* row = s + (int64_t)blockIdx.x * lds;
    T* orow = out + (int64_t)blockIdx.x * ldo;
    auto block_max = [&](float v) {
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        if (lane == 0) red[wid] = v;
        __syncthreads();
        v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        return v;
    };
    auto block_sum = [&](float v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wid] = v;
        __syncthreads();
        v = (red[0] + red[1]) + (red[2] + red[3]);  // fixed order: bit-reproducible
        __syncthreads();
        return v;
    };
    constexpr int VPT = 16;  // float4 vectors per thread on the register path
    const bool vec = (lds % 4 == 0) && ((reinterpret_cast<uintptr_t>(s) & 15) == 0);
    if (L <= 256 * 4 * VPT && vec) {
        f32x4 v[VPT];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int col = (i * 256 + tid) * 4;
            if (col + 4 <= L) v[i] = *reinterpret_cast<const f32x4*>(row + col);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = col + e < L ? row[col + e] : -3.0e38f;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, v[i][e]);
        }
        mx = block_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = (i * 256 + tid) * 4 + e;
                v[i][e] = col < L ? fast_exp2((v[i][e] - mx) * c) : 0.f;
                sum += v[i][e];
            }
        sum = block_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = (i * 256 + tid) * 4 + e;
                if (col < Lp) orow[col] = from_f32<T>(v[i][e] * inv);
            }
        return;
    }
    float mx = -3.0e38f;
    for (int col = tid; col < L; col += 256) mx = fmaxf(mx, row[col]);
    mx = block_max(mx);
    float sum = 0.f;
    for (int col = tid; col < L; col += 256) sum += fast_exp2((row[col] - mx) * c);
    sum = block_sum(sum);
    const float inv = 1.0f / sum;
    for (int col = tid; col < Lp; col += 256) orow[col] = from_f32<T>(col < L ? fast_exp2((row[col] - mx) * c) * inv : 0.f);
}

// Self-Attention Guidance, mask side: acc[j] (+)= scale * sum_i P[i][j] -- the attention mass key j receives from all queries of one
// head (self_attention_guidance.py:80: attn_map.mean(heads).sum(queries)).  One workgroup per 64 columns, 4 row lanes, fixed-order
// LDS reduction; heads are accumulated by consecutive launches (accumulate = 1), so the result is deterministic.
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* __restrict__ p, int64_t ldp, int M, int L, float* __restrict__ acc, int accumulate, float scale) {
    __shared__ float red[256];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float s = 0.f;
    if (col < L)
        for (int i = rl; i < M; i += 4) s += to_f32(p[(int64_t)i * ldp + col]);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && col < L) {
        const float t = ((red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192])) * scale;
        acc[col] = accumulate ? acc[col] + t : t;
    }
}

// Self-Attention Guidance, latent side (SAGAdapter.compute_degraded_latents, self_attention_guidance.py:62-95) in one pass:
//   x0   = (x - noise_std * eps) / scale_factor                      solver.remove_noise
//   blur = gaussian_blur(x0, k x k, reflect padding)                 separable weights w1[k] from the host
//   m    = mass[b][nearest (ah, aw) cell of the pixel] > 1           compute_sag_mask + nearest interpolate
//   out  = scale_factor * (m ? blur : x0) + noise_std * eps          solver.add_noise
// x, eps, out: [n][C][h][w]; coef[1] = scale factor, coef[2] = noise std of the step (the CFG+DDIM kernel's device table row).
template <typename T>
__global__ __launch_bounds__(256) void sag_degrade_kernel(const T* __restrict__ x, const T* __restrict__ eps, const float* __restrict__ mass, int ah, int aw,
                                                           const float* __restrict__ coef, const float* __restrict__ w1, int ks, T* __restrict__ out, int n, int C, int h,
                                                           int w) {
    const int64_t total = (int64_t)n * C * h * w;
    const float a = coef[1], sd = coef[2];
    const int half = ks / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const int64_t plane = i / ((int64_t)h * w);  // b * C + c
        const int b = (int)(plane / C);
        const T* xp = x + plane * h * w;
        const T* ep = eps + plane * h * w;
        const float e0 = to_f32(ep[py * w + px]);
        const float x0 = (to_f32(xp[py * w + px]) - sd * e0) / a;
        const int cy = (int)(((int64_t)py * ah) / h), cx = (int)(((int64_t)px * aw) / w);
        float v = x0;
        if (mass[(int64_t)b * ah * aw + cy * aw + cx] > 1.0f) {
            float acc = 0.f;
            for (int dy = 0; dy < ks; ++dy) {
                int yy = py + dy - half;
                yy = yy < 0 ? -yy : (yy >= h ? 2 * (h - 1) - yy : yy);
                float row = 0.f;
                for (int dx = 0; dx < ks; ++dx) {
                    int xx = px + dx - half;
                    xx = xx < 0 ? -xx : (xx >= w ? 2 * (w - 1) - xx : xx);
                    row += w1[dx] * ((to_f32(xp[yy * w + xx]) - sd * to_f32(ep[yy * w + xx])) / a);
                }
                acc += w1[dy] * row;
            }
            v = acc;
        }
        out[i] = from_f32<T>(a * v + sd * e0);
    }
}

#define DISPATCH_T(dtype, CALL)                          \
    do {                                                 \
        if ((dtype) == MI355X_F32) {                     \
            using T = float;                             \
            CALL;                                        \
        } else if ((dtype) == MI355X_BF16) {             \
            using T = bf16_t;                            \
            CALL;                                        \
        } else                                           \
            return MI355X_EDTYPE;                        \
    } while (0)

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH)

extern "C" int mi355x_abi_version(void) { return MI355X_ABI_VERSION; }

extern "C" int mi355x_device_info(char* buf, int32_t buflen) {
    if (!buf || buflen <= 0) return MI355X_EARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MI355X_ELAUNCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MI355X_ELAUNCH;
    snprintf(buf, (size_t)buflen, "%s %s CUs=%d LDS/block=%zu clock=%dkHz mem=%zuMiB", prop.name, prop.gcnArchName,
             prop.multiProcessorCount, (size_t)prop.sharedMemPerBlock, prop.clockRate, (size_t)(prop.totalGlobalMem >> 20));
    return MI355X_OK;
}

extern "C" int mi355x_nchw_to_nhwc(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t HW, int64_t ldo, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0 || ldo < C) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
    DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), grid, dim3(256), 0, st, static_cast<const T*>(x), static_cast<T*>(out), C, HW, ldo));
    return LAUNCH_OK();
}

extern "C" int mi355x_nhwc_to_nchw(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t HW, int64_t ldx, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0 || ldx < C) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
    DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), grid, dim3(256), 0, st, static_cast<const T*>(x), static_cast<T*>(out), C, HW, ldx));
    return LAUNCH_OK();
}

extern "C" int mi355x_im2col3x3_nchw(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int64_t ldo,
                                      void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ldo < 9 * C) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((int64_t)B * H * W * ldo);
    DISPATCH_T(dtype, hipLaunchKernelGGL((im2col3x3_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<T*>(out), B, C, H, W, ldo));
    return LAUNCH_OK();
}

extern "C" int mi355x_concat2(int32_t dtype, const void* a, int64_t lda, int32_t C1, const void* b, int64_t ldb, int32_t C2, void* out,
                               int64_t ldo, int64_t M, void* stream) {
    if (!a || !b || !out || M <= 0 || C1 <= 0 || C2 <= 0) return MI355X_EARG;
    if (dtype != MI355X_F32 && dtype != MI355X_BF16) return MI355X_EDTYPE;
    const int es = dtype == MI355X_F32 ? 4 : 2;
    if ((C1 * es) % 16 || (C2 * es) % 16 || (lda * es) % 16 || (ldb * es) % 16 || (ldo * es) % 16 || !al16(a) || !al16(b) || !al16(out))
        return MI355X_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int epc = 16 / es;
    const int grid = grid_for(M * ((C1 + C2) / epc));
    DISPATCH_T(dtype, hipLaunchKernelGGL((concat2_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(a), lda, C1 / epc,
                                         static_cast<const T*>(b), ldb, C2 / epc, static_cast<T*>(out), ldo, M));
    return LAUNCH_OK();
}

extern "C" int mi355x_axpby(int32_t dtype, const void* a, float alpha, const void* b, float beta, void* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return MI355X_EARG;
    if (!al16(a) || !al16(b) || !al16(out)) return MI355X_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n / 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(a), alpha,
                                         static_cast<const T*>(b), beta, static_cast<T*>(out), n));
    return LAUNCH_OK();
}

extern "C" int mi355x_silu(int32_t dtype, const void* x, void* out, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n);
    DISPATCH_T(dtype, hipLaunchKernelGGL((silu_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<T*>(out), n));
    return LAUNCH_OK();
}

extern "C" int mi355x_cfg_ddim_step(int32_t dtype, void* x, const void* unet_out, const float* coef, int64_t n, void* stream) {
    if (!x || !unet_out || !coef || n <= 0) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n);
    DISPATCH_T(dtype, hipLaunchKernelGGL((cfg_ddim_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<T*>(x), static_cast<const T*>(unet_out), coef, n));
    return LAUNCH_OK();
}

extern "C" int mi355x_cfg_linear_step(int32_t dtype, void* x, const void* unet_out, void* hist, void* model_in, const float* coef, int64_t n,
                                      void* stream) {
    if (!x || !unet_out || !hist || !coef || n <= 0) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n);
    DISPATCH_T(dtype, hipLaunchKernelGGL((cfg_linear_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<T*>(x), static_cast<const T*>(unet_out),
                                         static_cast<T*>(hist), static_cast<T*>(model_in), coef, n));
    return LAUNCH_OK();
}

extern "C" int mi355x_sinusoidal(int32_t dtype, const float* x, int64_t n, int32_t dim, int32_t group, void* out, int64_t ldo, int32_t col0,
                                 void* stream) {
    if (!x || !out || n <= 0 || dim <= 0 || dim % 2 || group <= 0) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for(n * (dim / 2));
    DISPATCH_T(dtype, hipLaunchKernelGGL((sinusoidal_kernel<T>), dim3(grid), dim3(256), 0, st, x, n, (int)dim, (int)group, static_cast<T*>(out), ldo, (int)col0));
    return LAUNCH_OK();
}

extern "C" int mi355x_patchify_nchw(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t P, int64_t ldo,
                                    void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || P <= 0 || H % P || W % P) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((int64_t)B * (H / P) * (W / P) * C * P * P);
    DISPATCH_T(dtype, hipLaunchKernelGGL((patchify_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<T*>(out), B, C, H, W, P, ldo));
    return LAUNCH_OK();
}

extern "C" int mi355x_gather_rows(int32_t dtype, const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int64_t n_rows, int32_t C,
                                  void* stream) {
    if (!x || !idx || !out || n_rows <= 0 || C <= 0) return MI355X_EARG;
    const int es = dtype == MI355X_F32 ? 4 : 2;
    if ((C * es) % 16 || (ldx * es) % 16 || (ldo * es) % 16 || !al16(x) || !al16(out)) return MI355X_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nv = C * es / 16;
    const int grid = grid_for(n_rows * nv);
    DISPATCH_T(dtype, hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), ldx, idx, static_cast<T*>(out), ldo, n_rows, nv));
    return LAUNCH_OK();
}

extern "C" int mi355x_pointwise_nchw(int32_t dtype, const void* x, const void* w, const void* bias, void* out, int32_t B, int32_t Ci, int32_t Co,
                                     int64_t HW, void* stream) {
    if (!x || !w || !out || B <= 0 || HW <= 0 || Ci <= 0 || Co <= 0 || Ci > 8 || Co > 8) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = grid_for((int64_t)B * HW);
    DISPATCH_T(dtype, hipLaunchKernelGGL((pointwise_nchw_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<const T*>(w),
                                         static_cast<const T*>(bias), static_cast<T*>(out), B, Ci, Co, HW));
    return LAUNCH_OK();
}

// ---- SegmentAnything decomposed relative position bias -> extra query columns --------------------------------------------
// src row (one token), per head: [ q * scale (d) | P1 (2*S1 - 1) | P2 (2*S2 - 1) | pad ]  (Lp columns), where
// P1[r] = q . E1[2*S1 - 2 - r] and P2[r] = q . E2[2*S2 - 2 - r] come out of the same GEMM as q (embedding tables folded into
// the projection weights, reversed).  Token t of a sample sits at grid position (a, b) = (t / S2, t % S2); its bias against a key
// at (a', b') is q . E1[a - a' + S1 - 1] + q . E2[b - b' + S2 - 1] = P1[S1 - 1 - a + a'] + P2[S2 - 1 - b + b'], i.e. two
// CONTIGUOUS windows of the P columns.  out row per head: [ q * scale (d) | P1 window (S1) | P2 window (S2) | 0 pad ] (Dq columns)
// which mi355x_attention_general multiplies with K' = [ k | onehot(a') | onehot(b') | 0 ].
// One work item = one 16-byte vector of an output row (EPC elements of one head): 32-bit index arithmetic, one vector store per item.  The q part of a head
// (columns [0, d)) is copied with vector loads where source and destination are 16-byte aligned; the two P windows start at a per-token offset of the source, so
// their elements are gathered one by one (S1 + S2 of the Dq columns).  (Round 6: the first version handled one ELEMENT per item with 64-bit divisions -- 28.8 us per
// launch for 17.6 MB of output in the SAM encoder's windowed blocks, 0.92 ms of its 12.7.)
template <typename T>
__global__ __launch_bounds__(256) void relpos_pack_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ out, int64_t ldo, int64_t M, int H, int d,
                                                           int S1, int S2, int Lp, int Dq, int vec) {
    constexpr int EPC = DT<T>::EPC;
    const int L = S1 * S2;
    if (vec) {
        const unsigned nv = (unsigned)Dq / EPC, per_row = nv * (unsigned)H;
        const uint64_t total = (uint64_t)M * per_row;
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
            const unsigned row = (unsigned)(i / per_row), rem = (unsigned)(i - (uint64_t)row * per_row);  // (M < 2^32 rows)
            const unsigned hd = rem / nv, v = rem - hd * nv;
            const int col0 = (int)v * EPC;
            const T* sp = src + (int64_t)row * lds + (int64_t)hd * Lp;
            T* op = out + (int64_t)row * ldo + (int64_t)hd * Dq + col0;
            if (col0 + EPC <= d) {
                store16<T>(op, load16<T>(sp + col0));
                continue;
            }
            const int t = (int)(row % (unsigned)L), a = t / S2, b = t - a * S2;
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int col = col0 + e;
                float x = 0.f;
                if (col < d) x = to_f32(sp[col]);
                else if (col < d + S1) x = to_f32(sp[d + (S1 - 1 - a) + (col - d)]);
                else if (col < d + S1 + S2) x = to_f32(sp[d + (2 * S1 - 1) + (S2 - 1 - b) + (col - d - S1)]);
                ov.set(e, x);
            }
            store16<T>(op, ov);
        }
        return;
    }
    const int64_t total = M * H * Dq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / ((int64_t)H * Dq);
        const int rem = (int)(i - row * H * Dq);
        const int hd = rem / Dq, col = rem - hd * Dq;
        const int t = (int)(row % L), a = t / S2, b = t - a * S2;
        const T* s = src + row * lds + (int64_t)hd * Lp;
        T v;
        if (col < d) v = s[col];
        else if (col < d + S1) v = s[d + (S1 - 1 - a) + (col - d)];
        else if (col < d + S1 + S2) v = s[d + (2 * S1 - 1) + (S2 - 1 - b) + (col - d - S1)];
        else v = from_f32<T>(0.f);
        out[row * ldo + rem] = v;
    }
}

extern "C" int mi355x_relpos_pack(int32_t dtype, const void* src, int64_t lds, void* out, int64_t ldo, int64_t M, int32_t H, int32_t d, int32_t S1,
                                  int32_t S2, int32_t Lp, int32_t Dq, void* stream) {
    if (!src || !out || M <= 0 || H <= 0 || d <= 0 || S1 <= 0 || S2 <= 0) return MI355X_EARG;
    if (Lp < d + 2 * S1 - 1 + 2 * S2 - 1 || Dq < d + S1 + S2 || lds < (int64_t)H * Lp || ldo < (int64_t)H * Dq) return MI355X_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int es = dtype == MI355X_F32 ? 4 : 2, epc = 16 / es;
    // the vector path: every head's slice of both rows starts on a 16-byte boundary and holds whole vectors
    const int vec = (reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && (lds * es) % 16 == 0 && (ldo * es) % 16 == 0 && Lp % epc == 0 &&
                     Dq % epc == 0 && d % epc == 0 && M < (1ll << 32)) ? 1 : 0;
    const int grid = grid_for(vec ? M * H * (Dq / epc) : M * H * Dq);
    DISPATCH_T(dtype, hipLaunchKernelGGL((relpos_pack_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(src), lds, static_cast<T*>(out), ldo, M, H, d,
                                         S1, S2, Lp, Dq, vec));
    return LAUNCH_OK();
}

extern "C" int mi355x_softmax_rows(int32_t dtype, const float* s, int64_t lds, void* out, int64_t ldo, int64_t M, int32_t L, int32_t Lp, float scale,
                                   void* stream) {
    if (!s || !out || M <= 0 || L <= 0 || Lp < L || lds < L || ldo < Lp) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float c = scale * 1.44269504088896340736f;
    DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)M), dim3(256), 0, st, s, lds, static_cast<T*>(out), ldo, L, Lp, c));
    return LAUNCH_OK();
}

extern "C" int mi355x_colsum_rows(int32_t dtype, const void* p, int64_t ldp, int32_t M, int32_t L, float* acc, int32_t accumulate, float scale, void* stream) {
    if (!p || !acc || M <= 0 || L <= 0 || ldp < L) return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_rows_kernel<T>), dim3((L + 63) / 64), dim3(256), 0, st, static_cast<const T*>(p), ldp, M, L, acc, accumulate, scale));
    return LAUNCH_OK();
}

extern "C" int mi355x_sag_degrade(int32_t dtype, const void* x, const void* eps, const float* mass, int32_t ah, int32_t aw, const float* coef, const float* w1,
                                  int32_t ksize, void* out, int32_t n, int32_t C, int32_t h, int32_t w, void* stream) {
    if (!x || !eps || !mass || !coef || !w1 || !out || n <= 0 || C <= 0 || h <= 0 || w <= 0 || ah <= 0 || aw <= 0 || ksize < 1 || !(ksize & 1) || ksize / 2 >= h || ksize / 2 >= w)
        return MI355X_EARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)n * C * h * w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DISPATCH_T(dtype, hipLaunchKernelGGL((sag_degrade_kernel<T>), dim3((int)blocks), dim3(256), 0, st, static_cast<const T*>(x), static_cast<const T*>(eps), mass, ah, aw, coef, w1,
                                         ksize, static_cast<T*>(out), n, C, h, w));
    return LAUNCH_OK();
}
