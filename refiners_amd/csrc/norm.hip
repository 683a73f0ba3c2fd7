// LayerNorm and NHWC GroupNorm(+SiLU) for gfx950: HBM-bound streaming kernels, 16-byte vectorised, deterministic.
//
// LayerNorm: one wave per row, the row lives in registers (two-pass mean / variance like F.layer_norm), 4 rows per
// workgroup.  GroupNorm works on the NHWC activations the conv / attention kernels use (no NCHW detour):
//   1. gn_partial : each workgroup reduces a pixel range of one sample into per-CHANNEL shifted sums (pivot = the
//                   channel's value at pixel 0, which removes the E[x^2]-E[x]^2 cancellation), fixed order;
//   2. gn_finalize: per sample, channel partials -> (mean_c, M2_c) -> Chan merge into the 32 groups -> per-channel
//                   (mean_g, rstd_g * gamma_c) table;
//   3. gn_apply   : y = x * a + (beta - mean * a), optional SiLU, one read + one write of the activation, (a, shift) held in registers.
#include "common.cuh"
#include "../../include/mi355x_refiners.h"

int g_gn_wgs = 2048;  // workgroups a GroupNorm pass aims for (512 until round 5: profiles/r05_r_probe_gn_apply.log, -5 ... -15 % at 4 images per GPU, nil at a CFG pair); (mi355x_set_option "gnwgs", before any workspace is sized: mi355x_groupnorm_ws_floats follows it)
int g_gn_unroll = 4;  // loads in flight per thread of the apply pass: 4 or 8 ("gnunroll")

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ gamma,
                                                         const T* __restrict__ beta, T* __restrict__ out, int64_t ldo, int M,
                                                         int C, float eps) {
    constexpr int EPC = DT<T>::EPC;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    if (row >= M) return;
    const int nvec = C / EPC;
    const T* xp = x + (int64_t)row * ldx;
    float v[MAXV][EPC];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec) {
            Vec16<T> t = load16<T>(xp + vi * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                v[k][e] = t.get(e);
                s += v[k][e];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d = v[k][e] - mean;
                ss += d * d;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = rsqrtf(ss / (float)C + eps);
    T* op = out + (int64_t)row * ldo;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec) {
            Vec16<T> gm = load16<T>(gamma + vi * EPC), bt = load16<T>(beta + vi * EPC), o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, (v[k][e] - mean) * rstd * gm.get(e) + bt.get(e));
            store16<T>(op + vi * EPC, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GroupNorm
// ws layout (floats): part[B][nchunk][C][2] | tab[B][C][2]
constexpr int GN_MAXVPT = 4;

// Two-source form (every GroupNorm kernel below): channels [0, C1) come from x (pixel stride ldx), channels [C1, C) from x2 (ldx2) -- the
// ResidualConcatenator's output never exists as a tensor (unet.py:69-85: Concatenate(x, residuals[n]) feeding a ResidualBlock).  One source: C1 = C.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int64_t ldx1, const T* __restrict__ x2, int64_t ldx2, int C1, int HW, int C, int ppc,
                                                          int nchunk, float* __restrict__ part) {
    constexpr int EPC = DT<T>::EPC;
    __shared__ float red[256 * EPC * 2];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int NV = C / EPC;
    const int tid = threadIdx.x;
    const int PL = NV >= 256 ? 1 : 256 / NV;             // pixel lanes
    const int VPT = NV >= 256 ? (NV + 255) / 256 : 1;    // vectors per thread
    const int pl = NV >= 256 ? 0 : tid / NV;
    const int v0 = NV >= 256 ? tid : tid % NV;
    const bool active = pl < PL;
    const int p0 = chunk * ppc;
    const int p1 = min(p0 + ppc, HW);
    for (int k = 0; k < VPT; ++k) {
        const int v = v0 + 256 * k;
        const bool on = active && v < NV;
        float s1[EPC], s2[EPC], piv[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = piv[e] = 0.f;
        if (on) {
            const bool sec = v * EPC >= C1;
            const int64_t ldx = sec ? ldx2 : ldx1;
            const T* xv = sec ? x2 + (int64_t)b * HW * ldx2 + (v * EPC - C1) : x + (int64_t)b * HW * ldx1 + v * EPC;
            Vec16<T> pv = load16<T>(xv);  // pixel 0 of this sample = pivot
#pragma unroll
            for (int e = 0; e < EPC; ++e) piv[e] = pv.get(e);
            int px = p0 + pl;
            for (; px + 3 * PL < p1; px += 4 * PL) {  // four independent 16-byte loads in flight (a one-load loop runs at one memory round trip per pixel)
                Vec16<T> t0 = load16<T>(xv + (int64_t)px * ldx), t1 = load16<T>(xv + (int64_t)(px + PL) * ldx), t2 = load16<T>(xv + (int64_t)(px + 2 * PL) * ldx),
                         t3 = load16<T>(xv + (int64_t)(px + 3 * PL) * ldx);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d0 = t0.get(e) - piv[e], d1 = t1.get(e) - piv[e], d2 = t2.get(e) - piv[e], d3 = t3.get(e) - piv[e];
                    s1[e] += (d0 + d1) + (d2 + d3);
                    s2[e] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
            for (; px < p1; px += PL) {
                Vec16<T> t = load16<T>(xv + (int64_t)px * ldx);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d = t.get(e) - piv[e];
                    s1[e] += d;
                    s2[e] += d * d;
                }
            }
        }
        float* dst = part + (((int64_t)b * nchunk + chunk) * C) * 2;
        if (PL > 1) {
            // reduce over pixel lanes through LDS in a fixed order
            __syncthreads();
            if (on) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    red[(tid * EPC + e) * 2 + 0] = s1[e];
                    red[(tid * EPC + e) * 2 + 1] = s2[e];
                }
            }
            __syncthreads();
            if (on && pl == 0) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float a1 = 0.f, a2 = 0.f;
                    for (int q = 0; q < PL; ++q) {
                        a1 += red[((q * NV + v) * EPC + e) * 2 + 0];
                        a2 += red[((q * NV + v) * EPC + e) * 2 + 1];
                    }
                    dst[(v * EPC + e) * 2 + 0] = a1;
                    dst[(v * EPC + e) * 2 + 1] = a2;
                }
            }
        } else if (on) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                dst[(v * EPC + e) * 2 + 0] = s1[e];
                dst[(v * EPC + e) * 2 + 1] = s2[e];
            }
        }
    }
}

// One workgroup per (group, sample): the cg channels of the group x nchunk partials are reduced by 256 threads in a
// fixed order (L = 256 / cg lanes per channel, then a serial sum of the L lane totals), so the result is deterministic.
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ x2, int64_t ldx2, int C1, int HW, int C, int G,
                                                           int nchunk, const float* __restrict__ part,
                                                           const T* __restrict__ gamma, float eps, float* __restrict__ tab) {
    __shared__ double red[256 * 2];  // (double all the way: the partials are pivoted sums over up to HW / 4 pixels each, and S2 - S1^2 / n below is a difference of large numbers)
    __shared__ float mean_c[256], m2_c[256];
    __shared__ float stat[2];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = C / G;
    const int L = 256 / cg;  // lanes per channel (cg <= 256 is checked by the host)
    const int t = threadIdx.x;
    const int j = t / cg, cl = t - j * cg;  // consecutive threads -> consecutive channels: each k reads one contiguous run
    const bool on = j < L;
    const int c = g * cg + cl;
    const float n = (float)HW;
    double s1 = 0.0, s2 = 0.0;
    if (on) {
        const f32x2* pp = reinterpret_cast<const f32x2*>(part) + ((int64_t)b * nchunk * C + c);
        int k = j;
        for (; k + 7 * L < nchunk; k += 8 * L) {  // eight independent loads in flight, summed in a fixed tree
            f32x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(int64_t)(k + u * L) * C];
            s1 += (((double)v[0][0] + (double)v[1][0]) + ((double)v[2][0] + (double)v[3][0])) + (((double)v[4][0] + (double)v[5][0]) + ((double)v[6][0] + (double)v[7][0]));
            s2 += (((double)v[0][1] + (double)v[1][1]) + ((double)v[2][1] + (double)v[3][1])) + (((double)v[4][1] + (double)v[5][1]) + ((double)v[6][1] + (double)v[7][1]));
        }
        for (; k < nchunk; k += L) {
            const f32x2 v = pp[(int64_t)k * C];
            s1 += v[0];
            s2 += v[1];
        }
    }
    red[t * 2 + 0] = s1;
    red[t * 2 + 1] = s2;
    __syncthreads();
    if (on && j == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int q = 0; q < L; ++q) {
            a1 += red[(cl + cg * q) * 2 + 0];
            a2 += red[(cl + cg * q) * 2 + 1];
        }
        const float piv = to_f32(c < C1 ? x[(int64_t)b * HW * ldx + c] : x2[(int64_t)b * HW * ldx2 + (c - C1)]);
        mean_c[cl] = piv + (float)(a1 / (double)n);
        m2_c[cl] = fmaxf((float)(a2 - a1 * a1 / (double)n), 0.f);
    }
    __syncthreads();
    if (t == 0) {
        float mg = 0.f;
        for (int q = 0; q < cg; ++q) mg += mean_c[q];
        mg /= (float)cg;
        float m2 = 0.f;
        for (int q = 0; q < cg; ++q) {
            const float d = mean_c[q] - mg;
            m2 += m2_c[q] + n * d * d;
        }
        stat[0] = mg;
        stat[1] = rsqrtf(m2 / (n * (float)cg) + eps);
    }
    __syncthreads();
    if (t < cg) {
        float* tb = tab + ((int64_t)b * C + g * cg + t) * 2;
        tb[0] = stat[0];
        tb[1] = stat[1] * to_f32(gamma[g * cg + t]);
    }
}

// The same table from the PRODUCER's column statistics (mi355x_gemm_args.colstats_out: (sum, sum of squares) per 32-pixel block and channel,
// written by the epilogue of the convolution / GEMM that produced x): no pass over x at all.  One workgroup per (group, sample); thread t
// takes channel t % cg and the blocks t / cg, t / cg + L, ... (eight independent loads in flight), fixed-order tree; the raw moments are
// turned into (mean, M2) per channel in double (the subtraction S2 - S1^2 / n is where float32 would lose digits), groups merged like above.
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_cs_kernel(const float* __restrict__ cs, const float* __restrict__ cs2, int C1, int HW, int C, int G,
                                                              const T* __restrict__ gamma, float eps, float* __restrict__ tab) {
    __shared__ double red[256 * 2];
    __shared__ double mean_c[256];
    __shared__ float stat[2];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = C / G, nblk = HW / 32;
    const int L = 256 / cg;
    const int t = threadIdx.x;
    const int j = t / cg, cl = t - j * cg;
    const bool on = j < L;
    const int c = g * cg + cl;
    // The per-block moments are float32 (a GEMM epilogue wrote them); everything across blocks is summed in DOUBLE: S2 - S1^2 / n loses mean^2 / var
    // of its digits, and with float32 cross-block sums a channel of mean 100 and deviation 0.1 came out with a variance that was noise (round-4 advisor).
    double s1 = 0.0, s2 = 0.0;
    if (on) {
        // the two sources keep their own [block][channel] tables: C1 channels wide for x, C - C1 for x2
        const int Cs = c < C1 ? C1 : C - C1;
        const f32x2* pp = c < C1 ? reinterpret_cast<const f32x2*>(cs) + ((int64_t)b * nblk * C1 + c) : reinterpret_cast<const f32x2*>(cs2) + ((int64_t)b * nblk * Cs + (c - C1));
        // eight independent loads in flight in EVERY batch: the tail of a thread's blocks is padded with re-reads of its first block at weight zero
        // (round 6: a serial remainder loop -- the whole loop at 32 x 32 pixels, where 8 L exceeds the 32 blocks -- cost one L2 round trip per block)
        for (int k = j; k < nblk; k += 8 * L) {
            f32x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + u * L;
                v[u] = pp[(int64_t)(kk < nblk ? kk : k) * Cs];
                if (kk >= nblk) v[u] = f32x2{0.f, 0.f};
            }
            s1 += (((double)v[0][0] + (double)v[1][0]) + ((double)v[2][0] + (double)v[3][0])) + (((double)v[4][0] + (double)v[5][0]) + ((double)v[6][0] + (double)v[7][0]));
            s2 += (((double)v[0][1] + (double)v[1][1]) + ((double)v[2][1] + (double)v[3][1])) + (((double)v[4][1] + (double)v[5][1]) + ((double)v[6][1] + (double)v[7][1]));
        }
    }
    red[t * 2 + 0] = s1;
    red[t * 2 + 1] = s2;
    __syncthreads();
    const double n = (double)HW;
    double mc = 0.0, m2c = 0.0;  // threads 0 .. cg - 1: (mean, M2) of their channel
    if (on && j == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int q = 0; q < L; ++q) {
            a1 += red[(cl + cg * q) * 2 + 0];
            a2 += red[(cl + cg * q) * 2 + 1];
        }
        mc = a1 / n;
        const double m2 = a2 - a1 * a1 / n;
        m2c = m2 > 0.0 ? m2 : 0.0;
    }
    // the group's mean and M2 over its cg channels: two fixed-pairing tree reductions over 256 slots (deterministic; round 6: thread 0 walked the channels
    // twice on its own -- 2 x 80 dependent LDS round trips at 2 560 channels)
    auto block_sum = [&](double v) {
        __syncthreads();
        mean_c[t] = v;
        __syncthreads();
        for (int sft = 128; sft > 0; sft >>= 1) {
            if (t < sft) mean_c[t] += mean_c[t + sft];
            __syncthreads();
        }
        return mean_c[0];
    };
    const bool mine = on && j == 0;
    const double mg = block_sum(mine ? mc : 0.0) / (double)cg;
    const double dm = mc - mg;
    const double m2g = block_sum(mine ? m2c + n * dm * dm : 0.0);
    if (t == 0) {
        stat[0] = (float)mg;
        stat[1] = rsqrtf((float)(m2g / (n * (double)cg)) + eps);
    }
    __syncthreads();
    if (t < cg) {
        float* tb = tab + ((int64_t)b * C + g * cg + t) * 2;
        tb[0] = stat[0];
        tb[1] = stat[1] * to_f32(gamma[g * cg + t]);
    }
}

// One workgroup per pixel chunk of one sample (the chunks of gn_partial): a thread keeps the (scale, shift) of its 8 / 4 channels in registers
// and streams its pixels with four loads in flight -- per element one FMA (+ SiLU), no per-element table reads.
template <typename T, int U>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int64_t ldx1, const T* __restrict__ x2, int64_t ldx2, int C1, T* __restrict__ out, int64_t ldo,
                                                        int HW, int C, int ppc, const float* __restrict__ tab,
                                                        const T* __restrict__ beta, int silu) {
    constexpr int EPC = DT<T>::EPC;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int NV = C / EPC;
    const int tid = threadIdx.x;
    const int PL = NV >= 256 ? 1 : 256 / NV;
    const int VPT = NV >= 256 ? (NV + 255) / 256 : 1;
    const int pl = NV >= 256 ? 0 : tid / NV;
    const int v0 = NV >= 256 ? tid : tid % NV;
    if (pl >= PL) return;
    const int p0 = chunk * ppc;
    const int p1 = min(p0 + ppc, HW);
    for (int k = 0; k < VPT; ++k) {
        const int v = v0 + 256 * k;
        if (v >= NV) break;
        float sc[EPC], sh[EPC];
        {
            const f32x2* tb = reinterpret_cast<const f32x2*>(tab) + ((int64_t)b * C + v * EPC);
            Vec16<T> bt = load16<T>(beta + v * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const f32x2 ma = tb[e];
                sc[e] = ma[1];
                sh[e] = bt.get(e) - ma[0] * ma[1];
            }
        }
        const bool sec = v * EPC >= C1;
        const int64_t ldx = sec ? ldx2 : ldx1;
        const T* xv = sec ? x2 + (int64_t)b * HW * ldx2 + (v * EPC - C1) : x + (int64_t)b * HW * ldx1 + v * EPC;
        T* ov = out + (int64_t)b * HW * ldo + v * EPC;
        auto emit = [&](const Vec16<T>& t, int px) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float y = t.get(e) * sc[e] + sh[e];
                if (silu) y = silu_f(y);
                o.set(e, y);
            }
            store16<T>(ov + (int64_t)px * ldo, o);
        };
        int px = p0 + pl;
        for (; px + (U - 1) * PL < p1; px += U * PL) {  // U independent 16-byte loads in flight per thread
            Vec16<T> t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = load16<T>(xv + (int64_t)(px + u * PL) * ldx);
#pragma unroll
            for (int u = 0; u < U; ++u) emit(t[u], px + u * PL);
        }
        for (; px < p1; px += PL) emit(load16<T>(xv + (int64_t)px * ldx), px);
    }
}

inline int gn_ppc(int B, int HW, int C, int es) {
    const int nv = C * es / 16;
    const int pl = nv >= 256 ? 1 : 256 / nv;
    const int wgs = g_gn_wgs < 512 ? 512 : g_gn_wgs;
    int64_t target = ((int64_t)B * HW + wgs - 1) / wgs;  // ~wgs workgroups in total
    int ppc = (int)(target < pl ? pl : target);
    ppc = ((ppc + pl - 1) / pl) * pl;
    if (ppc < 4 * pl) ppc = 4 * pl;
    return ppc;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
int run_layernorm(const mi355x_layernorm_args* a, hipStream_t st) {
    constexpr int EPC = DT<T>::EPC;
    const int nvec = a->C / EPC;
    const int grid = (a->M + 3) / 4;
    const T* x = static_cast<const T*>(a->x);
    const T* gm = static_cast<const T*>(a->gamma);
    const T* bt = static_cast<const T*>(a->beta);
    T* o = static_cast<T*>(a->out);
    if (nvec <= 64) hipLaunchKernelGGL((layernorm_kernel<T, 1>), dim3(grid), dim3(256), 0, st, x, a->ldx, gm, bt, o, a->ldo, a->M, a->C, a->eps);
    else if (nvec <= 128) hipLaunchKernelGGL((layernorm_kernel<T, 2>), dim3(grid), dim3(256), 0, st, x, a->ldx, gm, bt, o, a->ldo, a->M, a->C, a->eps);
    else if (nvec <= 256) hipLaunchKernelGGL((layernorm_kernel<T, 4>), dim3(grid), dim3(256), 0, st, x, a->ldx, gm, bt, o, a->ldo, a->M, a->C, a->eps);
    else if (nvec <= 512) hipLaunchKernelGGL((layernorm_kernel<T, 8>), dim3(grid), dim3(256), 0, st, x, a->ldx, gm, bt, o, a->ldo, a->M, a->C, a->eps);
    else if (nvec <= 1024) hipLaunchKernelGGL((layernorm_kernel<T, 16>), dim3(grid), dim3(256), 0, st, x, a->ldx, gm, bt, o, a->ldo, a->M, a->C, a->eps);
    else return MI355X_ESHAPE;
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

template <typename T>
int run_groupnorm(const mi355x_groupnorm_args* a, hipStream_t st) {
    constexpr int EPC = DT<T>::EPC;
    const int es = sizeof(T);
    const int ppc = gn_ppc(a->B, a->HW, a->C, es);
    const int nchunk = (a->HW + ppc - 1) / ppc;
    float* part = a->ws;
    float* tab = a->ws + (int64_t)a->B * nchunk * a->C * 2;
    const T* x = static_cast<const T*>(a->x);
    const T* x2 = static_cast<const T*>(a->x2);
    const int C1 = x2 ? a->C1 : a->C;
    if (a->colstats) {  // the launch(es) that produced x (and x2) left their column statistics: no statistics pass over the tensor
        hipLaunchKernelGGL((gn_finalize_cs_kernel<T>), dim3(a->G, a->B), dim3(256), 0, st, a->colstats, a->colstats2, C1, a->HW, a->C, a->G, static_cast<const T*>(a->gamma), a->eps, tab);
    } else {
        hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(nchunk, a->B), dim3(256), 0, st, x, a->ldx, x2, a->ldx2, C1, a->HW, a->C, ppc, nchunk, part);
        hipLaunchKernelGGL((gn_finalize_kernel<T>), dim3(a->G, a->B), dim3(256), 0, st, x, a->ldx, x2, a->ldx2, C1, a->HW, a->C, a->G,
                           nchunk, part, static_cast<const T*>(a->gamma), a->eps, tab);
    }
    if (g_gn_unroll == 8)
        hipLaunchKernelGGL((gn_apply_kernel<T, 8>), dim3(nchunk, a->B), dim3(256), 0, st, x, a->ldx, x2, a->ldx2, C1, static_cast<T*>(a->out), a->ldo, a->HW, a->C, ppc, tab,
                           static_cast<const T*>(a->beta), a->silu);
    else
        hipLaunchKernelGGL((gn_apply_kernel<T, 4>), dim3(nchunk, a->B), dim3(256), 0, st, x, a->ldx, x2, a->ldx2, C1, static_cast<T*>(a->out), a->ldo, a->HW, a->C, ppc, tab,
                           static_cast<const T*>(a->beta), a->silu);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

}  // namespace

extern "C" int mi355x_layernorm(const mi355x_layernorm_args* a, void* stream) {
    if (!a || !a->x || !a->out || !a->gamma || !a->beta) return MI355X_EARG;
    if (a->dtype != MI355X_F32 && a->dtype != MI355X_BF16) return MI355X_EDTYPE;
    const int es = a->dtype == MI355X_F32 ? 4 : 2;
    if (a->M <= 0 || a->C <= 0 || (a->C * es) % 16 || (a->ldx * es) % 16 || (a->ldo * es) % 16) return MI355X_ESHAPE;
    if (!al16(a->x) || !al16(a->out) || !al16(a->gamma) || !al16(a->beta)) return MI355X_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return a->dtype == MI355X_F32 ? run_layernorm<float>(a, st) : run_layernorm<bf16_t>(a, st);
}

extern "C" int64_t mi355x_groupnorm_ws_floats(int32_t B, int32_t HW, int32_t C) {
    // worst case over both dtypes AND over every value of the "gnwgs" option (a probing switch that may be flipped after workspaces were sized:
    // round-5 advisor): the smallest pixels-per-chunk gn_ppc can return is 4 pixel lanes' worth, whatever the workgroup target
    int64_t best = 0;
    for (int es = 2; es <= 4; es += 2) {
        const int nv = C * es / 16;
        const int ppc = 4 * (nv >= 256 ? 1 : 256 / (nv > 0 ? nv : 1));
        const int64_t nchunk = (HW + ppc - 1) / ppc;
        const int64_t need = (int64_t)B * nchunk * C * 2 + (int64_t)B * C * 2;
        if (need > best) best = need;
    }
    return best;
}

extern "C" int mi355x_groupnorm(const mi355x_groupnorm_args* a, void* stream) {
    if (!a || !a->x || !a->out || !a->gamma || !a->beta || !a->ws) return MI355X_EARG;
    if (a->dtype != MI355X_F32 && a->dtype != MI355X_BF16) return MI355X_EDTYPE;
    const int es = a->dtype == MI355X_F32 ? 4 : 2;
    if (a->B <= 0 || a->HW <= 0 || a->C <= 0 || a->G <= 0 || a->C % a->G) return MI355X_ESHAPE;
    if ((a->C * es) % 16 || (a->ldx * es) % 16 || (a->ldo * es) % 16) return MI355X_ESHAPE;
    if ((a->C * es) / 16 > 256 * GN_MAXVPT || a->C / a->G > 256) return MI355X_ESHAPE;
    if (!al16(a->x) || !al16(a->out) || !al16(a->beta)) return MI355X_ESHAPE;
    if (a->colstats && (a->HW % 32 || (reinterpret_cast<uintptr_t>(a->colstats) & 7))) return MI355X_ESHAPE;
    if (a->x2) {  // second source: channels [C1, C)
        if (a->C1 <= 0 || a->C1 >= a->C || (a->C1 * es) % 16 || (a->ldx2 * es) % 16 || !al16(a->x2)) return MI355X_ESHAPE;
        if (a->colstats && (!a->colstats2 || (reinterpret_cast<uintptr_t>(a->colstats2) & 7))) return MI355X_ESHAPE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    return a->dtype == MI355X_F32 ? run_groupnorm<float>(a, st) : run_groupnorm<bf16_t>(a, st);
}
