// mi355x_attention, general head shapes: softmax(scale * Q K^T [causal]) V with a QK width Dqk and a V width Dv that need not
// be 64 and need not be equal.
//
// Who needs it (reference shapes): SD1.5's 8 heads over 320 / 640 / 1280 channels (head dims 40 / 80 / 160,
// stable_diffusion_1/unet.py:30-45), SegmentAnything ViT-H (16 heads of 80 whose decomposed relative-position bias is
// folded by the host into extra Q / K feature columns, so Dqk = 112 or 208 while Dv = 80,
// segment_anything/image_encoder.py:82-127), CLIP-style causal self-attention (is_causal, layers/attentions.py:60-202).
//
// Same wave-level scheme as attention.hip (S^T = K Q^T so a lane owns one query column, P^T already in B-operand layout,
// base-2 online softmax on raw v_exp_f32, 32 queries per wave, 4 waves, 64-key tiles, register-staged double-buffered LDS),
// generalised along the head dimension:
//   * QK^T runs over NS = ceil(Dqk / KSTEP) MMA steps; the K tile lives in LDS as NS planes of [64 keys][64 bytes]
//     (one MMA step's worth of columns per plane), 16-byte chunks XOR-swizzled by (row >> 2) & 3 so that a 16-row fragment
//     read touches 16 distinct 16-byte slots of a 256-byte bank row;
//   * P V runs over ND = ceil(Dv / 16) independent 16-row blocks of V^T, so Dv = 80 costs exactly 5 blocks (no padding to 128);
//   * columns / rows beyond Dqk / Dv are zero-filled by the loaders (predicated 16-byte loads), nothing is padded in HBM;
//   * `causal`: key j contributes to query i only if j <= i (both indices within the sample); tiles entirely above the
//     diagonal are skipped.
#include "common.cuh"
#include "../../include/mi355x_refiners.h"

namespace {

struct GAttnP {
    int B, H, Lq, Lk, Dqk, Dv, causal;
    const char* q;
    const char* k;
    const char* vt;
    char* out;
    int64_t ldqb, qbsb, ldkb, kbsb, ldvtb, vtbsb, ldob, obsb;  // bytes
    float c;          // scale * log2(e)
    float thr;        // FAST: a tile leaves the running maximum alone while no score exceeds it by more than thr = 8 / c
    float out_scale;
    int qtiles;
};

// lane <-> lane ^ 16 / lane ^ 32 as row / half swaps (gfx950; see attention.hip): the maximum over a query's four lane groups without an LDS round trip
MI_DEV float group_max4(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b), b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// FAST (bf16, round 6; the measures of attention.hip's self-attention loop that do not depend on its pipelining): lazy running maximum (cross-lane maximum and the
// O / l rescale only when some lane of the wave sees a score above reference + thr: P <= 2^8 otherwise, exact in the quotient), row sums l from the matrix pipe
// (one more MFMA per P^T fragment against a fragment of ones: l = sum of the ROUNDED P), permlane reductions instead of ds_bpermute.
template <typename T, int NS, int ND, bool FAST = false>
__global__ __launch_bounds__(256) void attn_general_kernel(const GAttnP p) {
    static_assert(!FAST || sizeof(T) == 2, "FAST is the bf16 path");
    constexpr int ES = sizeof(T);
    constexpr int EPC = DT<T>::EPC;
    constexpr int NW = 4, NTHR = 256, BQW = 32, BKV = 64;
    constexpr int VROWB = BKV * ES;           // V^T tile row: 64 keys
    constexpr int KPLANE = 64 * 64;           // one K plane: 64 keys x 64 bytes
    constexpr int KBYTES = NS * KPLANE;
    constexpr int VBYTES = ND * 16 * VROWB;
    constexpr int STAGE = KBYTES + VBYTES;
    constexpr int VCPR = VROWB / 16;          // 16-byte chunks per V^T row
    constexpr int VLI = (ND * 16 * VCPR + NTHR - 1) / NTHR;
    constexpr bool IS_BF16 = (ES == 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id();
    const int g = lane >> 4, c16 = lane & 15;
    int bid = blockIdx.x;
    const int qt = bid % p.qtiles;
    bid /= p.qtiles;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int q0 = qt * (BQW * NW) + wid * BQW;
    const int qk_chunks = (p.Dqk * ES + 15) / 16;  // valid 16-byte chunks of a Q / K row of this head
    const frag_t zero = frag_t{0, 0, 0, 0};

    // ---- Q fragments (B operand), predicated on the valid width ----
    frag_t qf[2][NS];
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        int qr = q0 + 16 * jq + c16;
        qr = qr < p.Lq ? qr : p.Lq - 1;
        const char* qp = p.q + (int64_t)b * p.qbsb + (int64_t)qr * p.ldqb + (int64_t)h * p.Dqk * ES;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ch = 4 * s + g;
            qf[jq][s] = ch < qk_chunks ? *reinterpret_cast<const frag_t*>(qp + ch * 16) : zero;
        }
    }

    const char* kbase = p.k + (int64_t)b * p.kbsb + (int64_t)h * p.Dqk * ES;
    const char* vbase = p.vt + (int64_t)h * p.Dv * p.ldvtb + (int64_t)b * p.vtbsb;
    const int krow = tid >> 2, kg = tid & 3;  // K loader: one 16-byte chunk of one key row per plane
    const int koff = krow * 64 + ((kg ^ ((krow >> 2) & 3)) << 4);
    frag_t kr[NS], vr[VLI];

    auto issue = [&](int tile) {
        const int kv0 = tile * BKV;
        int key = kv0 + (IS_BF16 ? k_row_key(krow) : krow);  // bf16: permuted key order (k_row_key, common.cuh)
        key = key < p.Lk ? key : p.Lk - 1;
        const char* kp = kbase + (int64_t)key * p.ldkb;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ch = 4 * s + kg;
            kr[s] = ch < qk_chunks ? *reinterpret_cast<const frag_t*>(kp + ch * 16) : zero;
        }
#pragma unroll
        for (int it = 0; it < VLI; ++it) {
            const int q = it * NTHR + tid, row = q / VCPR, pch = q % VCPR;
            vr[it] = (row < p.Dv) ? *reinterpret_cast<const frag_t*>(vbase + (int64_t)row * p.ldvtb + (int64_t)kv0 * ES + pch * 16) : zero;
        }
    };
    auto commit = [&](int buf) {
        char* ks = smem + buf * STAGE;
        char* vs = ks + KBYTES;
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<frag_t*>(ks + s * KPLANE + koff) = kr[s];
#pragma unroll
        for (int it = 0; it < VLI; ++it) {
            const int q = it * NTHR + tid, row = q / VCPR, pch = q % VCPR;
            if (row < ND * 16) *reinterpret_cast<frag_t*>(vs + tile_off<VROWB>(row, pch)) = vr[it];
        }
    };

    f32x4 o[ND][2];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-INFINITY, -INFINITY};
    float lsum[2] = {0.f, 0.f};
    f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // FAST: row sums as an MFMA accumulator (its four rows are equal)
    const frag_t ones = frag_t{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};

    int ntile = (p.Lk + BKV - 1) / BKV;
    if (p.causal) {  // keys beyond the last query of this workgroup never contribute
        const int last_q = min(qt * (BQW * NW) + BQW * NW - 1, p.Lq - 1);
        ntile = min(ntile, last_q / BKV + 1);
    }

    issue(0);
    commit(0);
    __syncthreads();

    for (int tile = 0; tile < ntile; ++tile) {
        const int cur = tile & 1;
        const bool more = tile + 1 < ntile;
        if (more) issue(tile + 1);
        const char* ks = smem + cur * STAGE;
        const char* vs = ks + KBYTES;
        const int kv0 = tile * BKV;

        // ---- S^T = K Q^T ----
        f32x4 st[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            st[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            st[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = 16 * t + c16;
            const int off = row * 64 + ((g ^ ((row >> 2) & 3)) << 4);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const frag_t kf = lds_read_frag(ks, s * KPLANE + off);
                mma_step<T>(st[t][0], kf, qf[0][s]);
                mma_step<T>(st[t][1], kf, qf[1][s]);
            }
        }
        // ---- masks: keys past Lk, and (causal) keys after the query ----
        if (kv0 + BKV > p.Lk || (p.causal && kv0 + BKV - 1 > q0)) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + (IS_BF16 ? k_row_key(16 * t + 4 * g + r) : 16 * t + 4 * g + r);
#pragma unroll
                    for (int jq = 0; jq < 2; ++jq) {
                        const int qi = q0 + 16 * jq + c16;
                        if (key >= p.Lk || (p.causal && key > qi)) st[t][jq][r] = -INFINITY;
                    }
                }
        }
        // ---- online softmax ----
        if constexpr (FAST) {
            float mloc[2];
            bool need = false;
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                float mx = st[0][jq][0];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][jq][r]);
                mloc[jq] = mx;
                need |= mx > mrun[jq] + p.thr;  // (reference still -inf: any finite score)
            }
            if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform
#pragma unroll
                for (int jq = 0; jq < 2; ++jq) {
                    const float mnew = fmaxf(mrun[jq], group_max4(mloc[jq]));
                    const float mref = mnew == -INFINITY ? 0.f : mnew;  // (a fully masked prefix: see below)
                    const float alpha = fast_exp2((mrun[jq] - mref) * p.c);
                    mrun[jq] = mnew;
                    lacc[jq] *= alpha;
#pragma unroll
                    for (int i = 0; i < ND; ++i) o[i][jq] *= alpha;
                }
            }
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                const float mc = (mrun[jq] == -INFINITY ? 0.f : mrun[jq]) * p.c;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[t][jq][r] = fast_exp2(st[t][jq][r] * p.c - mc);
            }
        }
#pragma unroll
        for (int jq = 0; jq < (FAST ? 0 : 2); ++jq) {
            float mx = st[0][jq][0];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][jq][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[jq], mx);
            // a fully masked prefix keeps mnew = -inf: use 0 as the reference point so that exp2(-inf - 0) = 0, not NaN
            const float mref = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = fast_exp2((mrun[jq] - mref) * p.c);
            const float mc = mref * p.c;
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(st[t][jq][r] * p.c - mc);
                    st[t][jq][r] = e;
                    ps += e;
                }
            lsum[jq] = lsum[jq] * alpha + ps;
            mrun[jq] = mnew;
#pragma unroll
            for (int i = 0; i < ND; ++i) o[i][jq] *= alpha;
        }
        // ---- O^T += V^T P^T ----
        if constexpr (IS_BF16) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                frag_t pb[2];
#pragma unroll
                for (int jq = 0; jq < 2; ++jq) {
                    bf16x8 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pk[r] = (bf16_t)st[2 * s2][jq][r];
                        pk[4 + r] = (bf16_t)st[2 * s2 + 1][jq][r];
                    }
                    pb[jq] = __builtin_bit_cast(frag_t, pk);
                }
                if constexpr (FAST) {
                    mma_step<T>(lacc[0], ones, pb[0]);
                    mma_step<T>(lacc[1], ones, pb[1]);
                }
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const frag_t vf = lds_read_frag(vs, tile_off<VROWB>(16 * i + c16, 4 * s2 + g));  // one chunk = the lane's 8 consecutive keys
                    mma_step<T>(o[i][0], vf, pb[0]);
                    mma_step<T>(o[i][1], vf, pb[1]);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const frag_t p0 = __builtin_bit_cast(frag_t, st[t][0]);
                const frag_t p1 = __builtin_bit_cast(frag_t, st[t][1]);
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    const frag_t vf = lds_read_frag(vs, tile_off<VROWB>(16 * i + c16, 4 * t + g));
                    mma_step<T>(o[i][0], vf, p0);
                    mma_step<T>(o[i][1], vf, p1);
                }
            }
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
    }

    // ---- store: lane owns d = 16 i + 4 g + r of query 16 jq + c16 ----
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int qr = q0 + 16 * jq + c16;
        if (qr >= p.Lq) continue;
        float l = lsum[jq];
        if constexpr (FAST) {
            l = lacc[jq][0];
        } else {
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = p.out_scale / l;
        T* op = reinterpret_cast<T*>(p.out + (int64_t)b * p.obsb + (int64_t)qr * p.ldob) + (int64_t)h * p.Dv;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = 16 * i + 4 * g;
            if (d0 + 4 <= p.Dv) {
                if constexpr (IS_BF16) {
                    bf16x4 v4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v4[r] = (bf16_t)(o[i][jq][r] * inv);
                    *reinterpret_cast<bf16x4*>(op + d0) = v4;
                } else {
                    *reinterpret_cast<f32x4*>(op + d0) = o[i][jq] * inv;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (d0 + r < p.Dv) op[d0 + r] = from_f32<T>(o[i][jq][r] * inv);
            }
        }
    }
}

int g_gattn_fast = 1;  // bf16 launches take the FAST instance (mi355x_attention_general_set_fast: A/B and tests)

template <typename T, int NS, int ND, bool FAST = false>
int launch_general(const GAttnP& p0, hipStream_t stream) {
    if constexpr (sizeof(T) == 2 && !FAST) {
        if (g_gattn_fast) return launch_general<T, NS, ND, true>(p0, stream);
    }
    constexpr int ES = sizeof(T);
    constexpr int LDS = 2 * (NS * 64 * 64 + ND * 16 * 64 * ES);
    auto kfn = attn_general_kernel<T, NS, ND, FAST>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    GAttnP p = p0;
    p.qtiles = (p.Lq + 127) / 128;
    hipLaunchKernelGGL(kfn, dim3(p.qtiles * p.H * p.B), dim3(256), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

// (QK steps, PV blocks) instantiated: the smallest pair that covers the request is used.
//   bf16: step = 32 columns; f32: step = 16 columns.  PV block = 16 rows of V^T for both.
template <typename T>
int dispatch_general(const GAttnP& p, hipStream_t st) {
    constexpr int KSTEP = DT<T>::KSTEP;
    const int ns = (p.Dqk + KSTEP - 1) / KSTEP, nd = (p.Dv + 15) / 16;
#define TRY(NS_, ND_) \
    if (ns <= (NS_ * 32 / KSTEP) && nd <= ND_) return launch_general<T, NS_ * 32 / KSTEP, ND_>(p, st);
    TRY(2, 3)    // head dim 40 (SD1.5 at 320 channels)
    TRY(2, 4)    // head dim 64 with causal masking (CLIP)
    TRY(3, 5)    // head dim 80 (SD1.5 at 640 channels)
    TRY(4, 5)    // SAM windowed: 80 + 14 + 14 bias columns, V 80
    TRY(5, 10)   // head dim 160 (SD1.5 at 1280 channels)
    TRY(7, 5)    // SAM global: 80 + 64 + 64 bias columns, V 80
#undef TRY
    return MI355X_ESHAPE;
}

inline bool al16g(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int mi355x_attention_general_set_fast(int v) {  // probing / A-B only, not part of the stable contract
    g_gattn_fast = v ? 1 : 0;
    return MI355X_OK;
}

extern "C" int mi355x_attention_general(const mi355x_attn_general_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->vt || !a->out) return MI355X_EARG;
    if (a->dtype != MI355X_F32 && a->dtype != MI355X_BF16) return MI355X_EDTYPE;
    const int es = a->dtype == MI355X_F32 ? 4 : 2;
    if (a->B <= 0 || a->H <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->Dqk <= 0 || a->Dv <= 0) return MI355X_ESHAPE;
    // every head's Q / K row segment and every V^T row must start on a 16-byte boundary
    if ((a->Dqk * es) % 16 || (a->ldq * es) % 16 || (a->ldk * es) % 16 || (a->ldvt * es) % 16 || a->ldo % 4 || a->Dv % 4) return MI355X_ESHAPE;
    if ((a->q_batch_stride * es) % 16 || (a->k_batch_stride * es) % 16 || (a->vt_batch_stride * es) % 16 || a->o_batch_stride % 4) return MI355X_ESHAPE;
    if (!al16g(a->q) || !al16g(a->k) || !al16g(a->vt) || (reinterpret_cast<uintptr_t>(a->out) & (4 * es - 1))) return MI355X_ESHAPE;
    GAttnP p{};
    p.B = a->B, p.H = a->H, p.Lq = a->Lq, p.Lk = a->Lk, p.Dqk = a->Dqk, p.Dv = a->Dv, p.causal = a->causal ? 1 : 0;
    p.q = static_cast<const char*>(a->q), p.k = static_cast<const char*>(a->k), p.vt = static_cast<const char*>(a->vt), p.out = static_cast<char*>(a->out);
    p.ldqb = a->ldq * es, p.qbsb = a->q_batch_stride * es, p.ldkb = a->ldk * es, p.kbsb = a->k_batch_stride * es;
    p.ldvtb = a->ldvt * es, p.vtbsb = a->vt_batch_stride * es, p.ldob = a->ldo * es, p.obsb = a->o_batch_stride * es;
    p.c = a->scale * 1.44269504088896340736f;
    p.thr = p.c > 0.f ? 8.0f / p.c : 0.f;
    p.out_scale = a->out_scale;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return a->dtype == MI355X_F32 ? dispatch_general<float>(p, st) : dispatch_general<bf16_t>(p, st);
}
