// mi355x_attention: flash-style scaled-dot-product attention for gfx950, head_dim 64, optional second K/V stream.
//
//   out[b, q, h*D + :] = sum_s out_scale_s * softmax_k( scale * Q[b,q,h] . K_s[b,k,h] ) V_s[b,k,h]
//
// Design (MI355X-first, not a port of a warp-32 flash kernel):
//   * "swapped" orientation: the wave computes S^T = K Q^T (MFMA A = K rows from LDS, B = Q rows held in registers),
//     so every lane owns ONE query column (c16) and 16 of the 64 keys of a tile: the online-softmax max / sum are
//     15 local ops + 2 cross-group shuffles, the rescale factor is a per-lane scalar, and P^T is already in the
//     register layout the second MFMA wants as its B operand (no LDS round trip, no permutes);
//   * V arrives TRANSPOSED from the projection GEMM (the GEMM is run with its operands swapped, which costs
//     nothing), so the V^T tile is K-contiguous like every other operand: plain swizzled LDS rows, 16-byte
//     conflict-free reads, identical code for bf16 and f32 (no ds_read_tr needed); in bf16 the K tile's rows are loaded
//     in a permuted key order (k_row_key) so that the eight P^T values a lane holds for one MFMA step belong to eight
//     CONSECUTIVE keys, i.e. to one 16-byte chunk of a V^T row;
//   * the V^T tile rows are loaded in the permuted order R = 16j + 4a + b <-> d = 16a + 4j + b, so each lane ends up
//     with 16 CONSECUTIVE head-dim outputs per query: the O store is 32/64 contiguous bytes per lane;
//   * K and V^T tiles (64 keys) go global -> registers -> LDS (double buffered, one barrier per tile) with the loads of the next TWO
//     tiles in flight (two register sets, the compiler's counted vmcnt releases only the older one): a CFG pair's 1024-token
//     self-attention is 320 workgroups on 256 CUs, one wave per SIMD, and with a single tile in flight its 16-tile loop ran at
//     the latency of one Infinity-Cache round trip per tile.  (A global_load_lds variant with a 3-deep ring is kept for A/B.)
//     Q, K, V are read in place from the (B, L, H*D) projection outputs: no head split / merge copies (reference: attentions.py:177-202).
//   * the q-tiles of one (batch, head) are remapped onto ONE XCD (xcd_remap), so a head's K / V^T (256 KB at 1024 keys) is pulled
//     into one private L2 instead of eight.
//   * softmax in base 2 with the scale folded into one FMA; running max starts at -inf; keys beyond Lk are masked to
//     -inf in the last tile only; deterministic (no atomics).
#include "common.cuh"
#include "../../include/mi355x_refiners.h"
#include <type_traits>

namespace {

struct KvP {
    const char* k;
    const char* vt;
    int64_t ldkb, kbsb;    // bytes
    int64_t ldvtb, vtbsb;  // bytes
    int Lk;
    float out_scale;
};

struct AttnP {
    int B, H, Lq, nstream;
    const char* q;
    int64_t ldqb, qbsb;
    char* out;
    int64_t ldob, obsb;
    float c;  // scale * log2(e)
    float thr;  // OPT bit 2: a tile leaves the running maximum alone while no score exceeds it by more than thr (= 8 / c: P <= 2^8)
    KvP kv[2];
    int qtiles;
    int xcd;  // 1 = XCD-aware block order
};

// lane <-> lane ^ 16 and lane <-> lane ^ 32 exchanges as row / half swaps (gfx950): after swap(a = x, b = x), a and b hold the two partners'
// values in every lane.  Inline asm (the builtin form folds the combine of its two results away when both inputs are the same value);
// s_nop 1 = the two wait states a VALU write of an operand needs before the swap reads it.
MI_DEV void xswap16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
MI_DEV void xswap32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
template <bool PL> MI_DEV float group_max(float x) {
    if constexpr (PL) {
        float a = x, b = x;
        xswap16(a, b);
        a = fmaxf(a, b), b = a;
        xswap32(a, b);
        return fmaxf(a, b);
    } else {
        x = fmaxf(x, __shfl_xor(x, 16));
        return fmaxf(x, __shfl_xor(x, 32));
    }
}
template <bool PL> MI_DEV float group_sum(float x) {
    if constexpr (PL) {
        float a = x, b = x;
        xswap16(a, b);
        a = a + b, b = a;
        xswap32(a, b);
        return a + b;
    } else {
        x += __shfl_xor(x, 16);
        return x + __shfl_xor(x, 32);
    }
}

template <typename T, int NW, int NSTREAM, bool GLDS, int NJQ, int RD = 2, int OPT = 0, int ABL = 0, int KVS = 1>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? 2 : 1))) void attn_kernel(const AttnP p) {
    // RD = register sets of the register-staged loader = K/V tiles in flight (1 or 2)
    // OPT bit 0: the two cross-group reductions of the online softmax as v_permlane16/32_swap (VALU) instead of ds_bpermute round trips;
    //     bit 1: all K fragments of a tile read before its first MFMA, all V^T fragments before the softmax (one exposed LDS latency per phase)
    //     bit 2 (bf16): LAZY running maximum -- every lane compares the maximum of its OWN 16 scores with (running maximum + thr); only when some lane of the
    //         wave is above (wave-uniform branch) are the cross-lane maximum, the new running maximum and the O / l rescale done at all.  Otherwise the tile
    //         is exponentiated against the old reference (P <= 2^8, exact in the quotient O / l): per tile and 16-query group 8 cross-lane / rescale
    //         operations + 8 packed multiplies of O fall away -- the loop is bound by its vector instructions (34 v_exp + ~120 others against 32 MFMAs)
    //     bit 3 (bf16): the row sums l come out of the matrix pipe -- one more MFMA per P^T fragment against a fragment of ones (l = sum of the ROUNDED P the
    //         P V product uses) instead of 16 vector adds per 16-query group and tile; no cross-lane sum at the end
    // ABL (probing only, results are wrong): 1 = no K/V traffic after the first tile, 2 = no softmax, 4 = no P V product, 8 = no Q K^T product,
    //     16 = no output store, 32 = no Q load, 64 = no K/V load at all
    // KVS = 2: key-split workgroup for grids that leave the SIMDs with one wave each (a CFG pair's 1024-token self-attention): the NW waves
    //     are NW/2 query groups x 2 key groups; key group kg works on keys 32 kg .. 32 kg + 31 of every 64-key tile (half of the Q K^T, softmax
    //     and P V chain per tile, its own running max / sum / O), and the two partial results are merged once at the end through LDS
    //     (O = O0 2^(c (m0 - m)) + O1 2^(c (m1 - m)), same for the sums).  Twice the workgroups for the same queries, each wave's dependent
    //     chain half as long; the price is that a workgroup stages the whole K / V^T for 64 queries instead of 128.
    // NJQ = 16-query groups per wave (2 = 32 queries; 1 = 16 queries: twice the waves per query, for launches too small to fill the chip)
    constexpr int D = 64, BKV = 64, BQW = 16 * NJQ;
    constexpr int ES = sizeof(T);
    constexpr int ROWB = D * ES;  // K rows and V^T rows have the same byte length (D == BKV)
    constexpr int CPR = ROWB / 16;
    constexpr int NTHR = NW * 64;
    constexpr int TILEB = 64 * ROWB;
    constexpr int STAGE = 2 * TILEB;
    constexpr int NST = GLDS ? 3 : 2;  // LDS ring depth: the glds loader keeps two K/V tiles in flight (counted vmcnt)
    constexpr int PD = NST - 1;
    constexpr int LI = 64 * CPR / NTHR;  // loader iterations per tile
    constexpr int LPS = 2 * LI;          // global_load_lds instructions per thread per K/V tile pair
    constexpr int NS = D / DT<T>::KSTEP;  // MMA steps over head dim
    constexpr bool IS_BF16 = (ES == 2);
    static_assert(64 * CPR % NTHR == 0, "loader mismatch");
    static_assert(KVS == 1 || (KVS == 2 && NSTREAM == 1 && !GLDS && NW % 2 == 0), "key-split: one stream, register-staged loader");
    constexpr int QW = NW / KVS;   // query groups (waves along the queries)
    constexpr int TT = 4 / KVS;    // 16-key slices of a tile per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id();
    const int g = lane >> 4, c16 = lane & 15;
    // grid: x = q tile (fastest), then head, then batch
    int bid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qt = bid % p.qtiles;
    bid /= p.qtiles;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int qg = KVS == 1 ? wid : wid % QW, kg = KVS == 1 ? 0 : wid / QW;
    const int q0 = qt * (BQW * QW) + qg * BQW;

    // ---- Q fragments (B operand), straight from global ----
    frag_t qf[NJQ][NS];
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        int qr = q0 + 16 * jq + c16;
        qr = qr < p.Lq ? qr : p.Lq - 1;
        const char* qp = p.q + (int64_t)b * p.qbsb + (int64_t)qr * p.ldqb + (int64_t)h * ROWB;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (ABL & 32) qf[jq][s] = frag_t{lane, s, jq, 0x3c003c00};
            else qf[jq][s] = *reinterpret_cast<const frag_t*>(qp + (4 * s + g) * 16);
        }
    }

    // ---- loader coordinates ----
    int lrow[LI], lcoff[LI], vrowd[LI];
#pragma unroll
    for (int it = 0; it < LI; ++it) {
        const int q = it * NTHR + tid, row = q / CPR, pch = q % CPR;
        lrow[it] = IS_BF16 ? k_row_key(row) : row;  // the KEY this LDS row of the K tile holds
        lcoff[it] = (pch ^ swz<ROWB>(row)) << 4;
        const int j = row >> 4, a = (row >> 2) & 3, bb = row & 3;
        vrowd[it] = 16 * a + 4 * j + bb;  // head-dim index stored in LDS row `row` of the V^T tile
    }
    frag_t kr[RD][LI], vr[RD][LI];

    f32x4 res[4][NJQ];
    if constexpr (NSTREAM > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) res[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int sidx = 0; sidx < NSTREAM; ++sidx) {
        const KvP& kv = p.kv[sidx];
        const int Lk = kv.Lk;
        const int ntile = (Lk + BKV - 1) / BKV;
        const char* kbase = kv.k + (int64_t)b * kv.kbsb + (int64_t)h * ROWB;
        const char* vbase = kv.vt + (int64_t)h * D * kv.ldvtb + (int64_t)b * kv.vtbsb;

        auto issue = [&](int tile, int buf, auto rs) {  // rs = register set (register-staged loader only)
            constexpr int RS = decltype(rs)::value;
            char* ks = smem + buf * STAGE;
            char* vs = ks + TILEB;
            const int kv0 = tile * BKV;
#pragma unroll
            for (int it = 0; it < LI; ++it) {
                int kr_ = kv0 + lrow[it];
                kr_ = kr_ < Lk ? kr_ : Lk - 1;
                const char* src = kbase + (int64_t)kr_ * kv.ldkb + lcoff[it];
                if constexpr (GLDS) glds16(src, ks + (it * NTHR + wid * 64) * 16);
                else kr[RS][it] = *reinterpret_cast<const frag_t*>(src);
            }
#pragma unroll
            for (int it = 0; it < LI; ++it) {
                const char* src = vbase + (int64_t)vrowd[it] * kv.ldvtb + (int64_t)kv0 * ES + lcoff[it];
                if constexpr (GLDS) glds16(src, vs + (it * NTHR + wid * 64) * 16);
                else vr[RS][it] = *reinterpret_cast<const frag_t*>(src);
            }
        };
        auto commit = [&](int buf, auto rs) {
            constexpr int RS = decltype(rs)::value;
            if constexpr (!GLDS) {
                char* ks = smem + buf * STAGE;
                char* vs = ks + TILEB;
#pragma unroll
                for (int it = 0; it < LI; ++it) *reinterpret_cast<frag_t*>(ks + (it * NTHR + tid) * 16) = kr[RS][it];
#pragma unroll
                for (int it = 0; it < LI; ++it) *reinterpret_cast<frag_t*>(vs + (it * NTHR + tid) * 16) = vr[RS][it];
            }
        };
        using rs0_t = std::integral_constant<int, 0>;
        using rs1_t = std::integral_constant<int, RD - 1>;

        f32x4 o[4][NJQ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
        float mrun[NJQ], lsum[NJQ];
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) mrun[jq] = -INFINITY, lsum[jq] = 0.f;
        constexpr bool LAZY = IS_BF16 && (OPT & 4) != 0 && KVS == 1 && !(ABL & 2);
        constexpr bool ONES = IS_BF16 && (OPT & 8) != 0 && KVS == 1 && !(ABL & 6);
        f32x4 lacc[NJQ];  // ONES: row sums as an MFMA accumulator (its four rows are equal)
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) lacc[jq] = f32x4{0.f, 0.f, 0.f, 0.f};
        const frag_t ones = frag_t{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};

        // all waves must be done reading LDS of the previous stream before it is overwritten
        __syncthreads();
        if constexpr (GLDS) {
#pragma unroll
            for (int s0 = 0; s0 < PD; ++s0)
                if (s0 < ntile) issue(s0, s0, rs0_t{});
        } else {
            if constexpr (!(ABL & 64)) issue(0, 0, rs0_t{});
            if constexpr (RD == 2) {
                if (ntile > 1) issue(1, 1, rs1_t{});
            }
            if constexpr (!(ABL & 64)) commit(0, rs0_t{});  // waits for set 0 only
            __syncthreads();
        }

        // one K/V tile; `par` = tile & 1 as a type, so that the register set is a compile-time index
        // `always` = the loop guarantees tile + 2 < ntile: the refill is then unconditional, which is what lets the compiler's vmcnt at the
        // commit leave it in flight (behind a branch it has to assume the committed set is the youngest and drains everything)
        auto tile_body = [&](int tile, auto par, auto always) {
            constexpr int PAR = decltype(par)::value;
            constexpr bool ALWAYS = decltype(always)::value;
            using cur_set = std::integral_constant<int, (RD == 2 ? PAR : 0)>;      // free set: tile `tile` was committed from it
            using nxt_set = std::integral_constant<int, (RD == 2 ? 1 - PAR : 0)>;  // holds tile + 1 (in flight) when RD == 2
            int cur;
            const bool more = tile + 1 < ntile;
            if constexpr (GLDS) {
                // tile `tile` has landed (the younger PD-1 stay in flight) -> raw barrier -> refill the buffer read last iteration
                if (tile + PD <= ntile) wait_vm<(PD - 1) * LPS>();
                else wait_vm0();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (tile + PD < ntile) issue(tile + PD, (tile + PD) % NST, rs0_t{});
                cur = tile % NST;
            } else {
                cur = PAR;
                if constexpr (RD == 2) {
                    if (!(ABL & 1) && (ALWAYS || tile + 2 < ntile)) issue(tile + 2, cur, cur_set{});
                } else {
                    if (!(ABL & 1) && more) issue(tile + 1, cur ^ 1, cur_set{});
                }
            }
            const char* ks = smem + ((ABL & 1) ? 0 : cur) * STAGE;
            const char* vs = ks + TILEB;

            // ---- S^T = K Q^T ----
            f32x4 st[TT][NJQ];
            if constexpr (ABL & 8) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) st[t][jq] = __builtin_bit_cast(f32x4, qf[jq][t % NS]);
            } else if constexpr ((OPT & 2) && KVS == 1) {
                frag_t kf[4][NS];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int s = 0; s < NS; ++s) kf[t][s] = lds_read_frag(ks, tile_off<ROWB>(16 * t + c16, 4 * s + g));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) st[t][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(st[t][jq], kf[t][s], qf[jq][s]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) st[t][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const frag_t kf = lds_read_frag(ks, tile_off<ROWB>(16 * (kg * TT + t) + c16, 4 * s + g));
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(st[t][jq], kf, qf[jq][s]);
                    }
                }
            }
            // V^T fragments of this tile, requested before the softmax so that they land under its VALU work (bf16, OPT bit 1)
            constexpr bool VPRE = IS_BF16 && (OPT & 2) && !(ABL & 4) && KVS == 1;
            frag_t vpre[VPRE ? 2 : 1][VPRE ? 4 : 1];
            if constexpr (VPRE) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        vpre[s2][i] = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * s2 + g));
                    }
            }
            // ---- mask the tail tile ----
            const int kv0 = tile * BKV;
            if (kv0 + BKV > Lk) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int krow = 16 * (kg * TT + t) + 4 * g + r;
                        if (kv0 + (IS_BF16 ? k_row_key(krow) : krow) >= Lk) {
#pragma unroll
                            for (int jq = 0; jq < NJQ; ++jq) st[t][jq][r] = -INFINITY;
                        }
                    }
            }
            // ---- online softmax (per lane: one query column per jq) ----
            if constexpr (LAZY) {
                float mloc[NJQ];
                bool need = false;
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) {
                    float mx = st[0][jq][0];
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][jq][r]);
                    mloc[jq] = mx;
                    need |= mx > mrun[jq] + p.thr;  // first tile: running maximum = -inf
                }
                if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform: some query of this wave moves its reference
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) {
                        const float mx = group_max<(OPT & 1) != 0>(mloc[jq]);
                        const float mnew = fmaxf(mrun[jq], mx);
                        const float alpha = fast_exp2((mrun[jq] - mnew) * p.c);
                        mrun[jq] = mnew;
                        if constexpr (ONES) lacc[jq] *= alpha;
                        else lsum[jq] *= alpha;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i][jq] *= alpha;
                    }
                }
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) {
                    const float mc = mrun[jq] * p.c;
                    float ps = 0.f;
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = fast_exp2(st[t][jq][r] * p.c - mc);
                            st[t][jq][r] = e;
                            if constexpr (!ONES) ps += e;
                        }
                    if constexpr (!ONES) lsum[jq] += ps;
                }
            }
#pragma unroll
            for (int jq = 0; jq < (LAZY ? 0 : NJQ); ++jq) {
                if constexpr (ABL & 2) {
                    float ps = 0.f;
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) ps += st[t][jq][r];
                    lsum[jq] += ps;
                    continue;
                }
                float mx = st[0][jq][0];
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][jq][r]);
                mx = group_max<(OPT & 1) != 0>(mx);
                float mnew = fmaxf(mrun[jq], mx);
                if constexpr (KVS == 2) mnew = fmaxf(mnew, -3.0e38f);  // a key group can see nothing but masked keys (Lk <= 32): keep the arithmetic finite
                const float alpha = fast_exp2((mrun[jq] - mnew) * p.c);
                const float mc = mnew * p.c;
                float ps = 0.f;
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = fast_exp2(st[t][jq][r] * p.c - mc);
                        st[t][jq][r] = e;
                        if constexpr (!ONES) ps += e;
                    }
                if constexpr (ONES) lacc[jq] *= alpha;
                else lsum[jq] = lsum[jq] * alpha + ps;
                mrun[jq] = mnew;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i][jq] *= alpha;
            }
            // ---- O^T += V^T P^T ----
            if constexpr (ABL & 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) o[i][jq] += st[i % TT][jq];
            } else if constexpr (IS_BF16) {
#pragma unroll
                for (int s2l = 0; s2l < 2 / KVS; ++s2l) {
                    const int s2 = KVS == 1 ? s2l : kg;  // 32-key half of the tile (key-split: this wave's half)
                    frag_t pb[NJQ];
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) {
                        bf16x8 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pk[r] = (bf16_t)st[2 * s2l][jq][r];
                            pk[4 + r] = (bf16_t)st[2 * s2l + 1][jq][r];
                        }
                        pb[jq] = __builtin_bit_cast(frag_t, pk);
                    }
                    if constexpr (ONES) {
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(lacc[jq], ones, pb[jq]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        frag_t vf;
                        if constexpr (VPRE) {
                            vf = vpre[s2l][i];
                        } else {
                            vf = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * s2 + g));
                        }
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(o[i][jq], vf, pb[jq]);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const frag_t vf = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * (kg * TT + t) + g));
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(o[i][jq], vf, __builtin_bit_cast(frag_t, st[t][jq]));
                    }
                }
            }
            if constexpr (!GLDS) {
                if (!(ABL & 1) && more) commit(cur ^ 1, nxt_set{});  // the compiler's vmcnt covers exactly this set: tile + 2 stays in flight
                __syncthreads();
            }
        };
        {
            using P0 = std::integral_constant<int, 0>;
            using P1 = std::integral_constant<int, 1>;
            int tile = 0;
            if constexpr (!GLDS && RD == 2) {
                for (; tile + 3 < ntile; tile += 2) {
                    tile_body(tile, P0{}, std::true_type{});
                    tile_body(tile + 1, P1{}, std::true_type{});
                }
            }
            for (; tile < ntile; tile += 2) {
                tile_body(tile, P0{}, std::false_type{});
                if (tile + 1 < ntile) tile_body(tile + 1, P1{}, std::false_type{});
            }
        }

        // ---- key-split: fold key group 1's partial result into key group 0's (LDS is free: the loop ended on a barrier) ----
        if constexpr (KVS == 2) {
            constexpr int NV = 18 * NJQ;  // 16 O values + running max + running sum per 16-query group
            float* mg = reinterpret_cast<float*>(smem) + (qg * NV) * 64 + lane;  // [query group][value][lane]: conflict-free
            if (kg == 1) {
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mg[(jq * 18 + 4 * i + r) * 64] = o[i][jq][r];
                    mg[(jq * 18 + 16) * 64] = mrun[jq];
                    mg[(jq * 18 + 17) * 64] = lsum[jq];
                }
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) {
                    const float m1 = mg[(jq * 18 + 16) * 64], l1 = mg[(jq * 18 + 17) * 64];
                    const float m = fmaxf(mrun[jq], m1);
                    const float a0 = fast_exp2((mrun[jq] - m) * p.c), a1 = fast_exp2((m1 - m) * p.c);  // m1 = -inf (no tile seen): a1 = 0
                    lsum[jq] = lsum[jq] * a0 + l1 * a1;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[i][jq][r] = o[i][jq][r] * a0 + mg[(jq * 18 + 4 * i + r) * 64] * a1;
                }
            }
        }

        // ---- finish this stream ----
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) {
            float l;
            if constexpr (ONES) l = lacc[jq][0];
            else l = group_sum<(OPT & 1) != 0>(lsum[jq]);
            const float inv = kv.out_scale / l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (NSTREAM > 1) res[i][jq] += o[i][jq] * inv;
                else res[i][jq] = o[i][jq] * inv;
            }
        }
    }

    // ---- store: lane owns d = 16g + 4i + r (16 consecutive) of query 16jq + c16 ----
    constexpr int EPC = DT<T>::EPC;
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        const int qr = q0 + 16 * jq + c16;
        if (qr >= p.Lq || kg != 0) continue;
        if constexpr (ABL & 16) {
            if (res[0][jq][0] != 12345.678f) continue;
        }
        T* op = reinterpret_cast<T*>(p.out + (int64_t)b * p.obsb + (int64_t)qr * p.ldob + (int64_t)h * ROWB) + 16 * g;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * i + r] = res[i][jq][r];
#pragma unroll
        for (int c = 0; c < 16 / EPC; ++c) {
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < EPC; ++e) ov.set(e, v[c * EPC + e]);
            store16<T>(op + c * EPC, ov);
        }
    }
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
MI_DEV rsrc_t make_rsrc(const char* base, int64_t bytes) {  // wave-uniform by construction; readfirstlane makes that provable (otherwise every load sits in a waterfall loop)
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const int64_t nb = bytes < 0x7ffffff0 ? bytes : 0x7ffffff0;
    const int n = __builtin_amdgcn_readfirstlane((int)nb);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, n, 0x00020000);
}
// 16 bytes per lane, global -> LDS: lane i lands at lds_wave_base + 16 i; source = descriptor base + voff (per lane) + soff (scalar); bytes beyond the descriptor's range read as zero
MI_DEV void blds16(rsrc_t rs, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------- software-pipelined loop
// attn_kernel's loop is a chain: Q K^T (MFMA) -> softmax (vector) -> P V (MFMA), and measured per wave-tile its time is the SUM of the matrix
// pipe's and the vector unit's (32 MFMAs x 16 cycles + ~120-200 vector instructions; removing any part removes exactly its own time,
// profiles/r06_r_probe_attn_ablate.log, r06_zi_probe_attn_opt.log): co-resident waves do not fill each other's gaps.  Here ONE wave keeps both
// busy: K runs one tile ahead of V, and an iteration is
//     phase 1      Q K^T of tile i + 1            ||  exponentials of tile i, query group 0
//     phase 1 + j  P V of tile i, group j - 1     ||  exponentials of tile i, group j           (32-query waves: one such phase)
//     last phase   P V of tile i, last group      ||  the maxima of tile i + 1 (the lazy running maximum's check)
// with the order MFMA, a few vector instructions, MFMA ... pinned by sched_group_barrier; the rare rescale (a wave-uniform branch) sits between
// the iterations.  Lazy running maximum and row sums from the matrix pipe as in attn_kernel's OPT bits 2 / 3.  bf16, one K/V stream, head_dim 64.
//   LDS: K(j) in K buffer j & 1, V^T(j) in V buffer j & 1; iteration i reads K(i + 1) and V(i), and commits K(i + 2) (into the buffer K(i) left in
//   iteration i - 1) and V(i + 1) (into the one V(i - 1) left) from registers loaded at its start; one barrier per iteration.
// ABL (probing, wrong results): 1 = no exponentials, 2 = no Q K^T MFMAs, 4 = no P V MFMAs, 8 = no K/V loads and commits in the loop, 16 = no barrier in the loop, 32 = no maxima
// DMA: K / V^T tiles go global -> LDS directly (buffer_load ... lds, issued at the START of the iteration into the buffers the previous iteration left:
//      the one-tile lead of K means they are free a whole iteration before they are read -- no deeper ring, no staging registers, no ds_write)
// Registers: 32-query waves 182 (2 waves per SIMD), 16-query waves 108 (4).  Measured and dropped (profiles/r06_zn_probe_attn_occ.log): budgets squeezed to 168 / 96 registers
// (a few spilled dwords: level at best, 2x slower for the 16-query waves), V^T fragments requested together with the K fragments (level).
// FOLD: the exponent's subtraction rides in the Q K^T product -- Q is scaled by c = scale * log2(e) once (rounded to bf16 again: a score error of ~1e-3, below the
//      rounding of P), the accumulators of tile i + 1 START at -(reference maximum, scaled) instead of 0, and P = exp2(accumulator): one v_exp per score, no v_fma
//      (32 of ~116 vector instructions per tile of a 32-query wave).  The reference only ever moves between iterations; a move subtracts its step from the scores
//      already computed against the old reference (the rare path).
template <int NJQ, int SCHED, int ABL = 0, bool DMA = false, bool FOLD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_pipe_kernel(const AttnP p) {
    using T = bf16_t;
    constexpr int NW = 4, D = 64, BKV = 64, BQW = 16 * NJQ, ES = 2, ROWB = D * ES, CPR = ROWB / 16, NTHR = NW * 64, TILEB = 64 * ROWB;
    constexpr int LI = 64 * CPR / NTHR, NS = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1

    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id();
    const int g = lane >> 4, c16 = lane & 15;
    int bid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qt = bid % p.qtiles;
    bid /= p.qtiles;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int q0 = qt * (BQW * NW) + wid * BQW;

    frag_t qf[NJQ][NS];
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        int qr = q0 + 16 * jq + c16;
        qr = qr < p.Lq ? qr : p.Lq - 1;
        const char* qp = p.q + (int64_t)b * p.qbsb + (int64_t)qr * p.ldqb + (int64_t)h * ROWB;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[jq][s] = *reinterpret_cast<const frag_t*>(qp + (4 * s + g) * 16);
            if constexpr (FOLD) {
                bf16x8 qv = __builtin_bit_cast(bf16x8, qf[jq][s]);
#pragma unroll
                for (int e = 0; e < 8; ++e) qv[e] = (bf16_t)((float)qv[e] * p.c);
                qf[jq][s] = __builtin_bit_cast(frag_t, qv);
            }
        }
    }

    const KvP& kv = p.kv[0];
    const int Lk = kv.Lk;
    const int ntile = (Lk + BKV - 1) / BKV;
    const char* kbase = kv.k + (int64_t)b * kv.kbsb + (int64_t)h * ROWB;
    const char* vbase = kv.vt + (int64_t)h * D * kv.ldvtb + (int64_t)b * kv.vtbsb;
    char* const kbuf = smem;
    char* const vbuf = smem + 2 * TILEB;
    // loader: a tile's address = wave-uniform base (scalar registers) + a per-lane 32-bit offset that does not change along the keys, so a load costs the
    // loop no vector arithmetic; the last tile's K rows beyond Lk are clamped through a second offset set
    uint32_t koff[LI], koff_last[LI], voff[LI];
#pragma unroll
    for (int it = 0; it < LI; ++it) {
        const int q = it * NTHR + tid, row = q / CPR, pch = q % CPR;
        const int key = k_row_key(row);
        const int coff = (pch ^ swz<ROWB>(row)) << 4;
        const int last = Lk - 1 - (ntile - 1) * BKV;
        koff[it] = (uint32_t)(key * (int)kv.ldkb + coff);
        koff_last[it] = (uint32_t)((key < last ? key : last) * (int)kv.ldkb + coff);
        const int j = row >> 4, a = (row >> 2) & 3, bb = row & 3;
        voff[it] = (uint32_t)((16 * a + 4 * j + bb) * (int)kv.ldvtb + coff);
    }

    const rsrc_t krs = make_rsrc(kbase, (int64_t)(Lk - 1) * kv.ldkb + ROWB);
    const rsrc_t vrs = make_rsrc(vbase, (int64_t)(D - 1) * kv.ldvtb + (int64_t)ntile * BKV * ES);
    auto dma_k = [&](int tile, int buf) {
        const uint32_t soff = (uint32_t)(tile * BKV) * (uint32_t)kv.ldkb;
        const bool lastt = tile == ntile - 1;
#pragma unroll
        for (int it = 0; it < LI; ++it) blds16(krs, kbuf + buf * TILEB + (it * NTHR + wid * 64) * 16, lastt ? koff_last[it] : koff[it], soff);
    };
    auto dma_v = [&](int tile, int buf) {
        const uint32_t soff = (uint32_t)(tile * BKV * ES);
#pragma unroll
        for (int it = 0; it < LI; ++it) blds16(vrs, vbuf + buf * TILEB + (it * NTHR + wid * 64) * 16, voff[it], soff);
    };
    frag_t kr[LI], vr[LI];
    auto load_k = [&](int tile) {
        const char* base = kbase + (int64_t)tile * BKV * kv.ldkb;
        const bool lastt = tile == ntile - 1;
#pragma unroll
        for (int it = 0; it < LI; ++it) kr[it] = *reinterpret_cast<const frag_t*>(base + (lastt ? koff_last[it] : koff[it]));
    };
    auto load_v = [&](int tile) {
        const char* base = vbase + (int64_t)tile * BKV * ES;
#pragma unroll
        for (int it = 0; it < LI; ++it) vr[it] = *reinterpret_cast<const frag_t*>(base + voff[it]);
    };
    auto commit_k = [&](int buf) {
#pragma unroll
        for (int it = 0; it < LI; ++it) *reinterpret_cast<frag_t*>(kbuf + buf * TILEB + (it * NTHR + tid) * 16) = kr[it];
    };
    auto commit_v = [&](int buf) {
#pragma unroll
        for (int it = 0; it < LI; ++it) *reinterpret_cast<frag_t*>(vbuf + buf * TILEB + (it * NTHR + tid) * 16) = vr[it];
    };

    f32x4 o[4][NJQ], lacc[NJQ];
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        lacc[jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float mrun[NJQ];  // the reference maximum: raw score units, -inf before the first tile; FOLD: scaled units (c * score), 0 before the first tile
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) mrun[jq] = FOLD ? 0.f : -INFINITY;
    const frag_t ones = frag_t{0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};

    auto qk = [&](const char* ks, f32x4(&sn)[4][NJQ]) {
        frag_t kf[4][NS];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s) kf[t][s] = lds_read_frag(ks, tile_off<ROWB>(16 * t + c16, 4 * s + g));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) {
                const float c0 = FOLD ? -mrun[jq] : 0.f;
                sn[t][jq] = f32x4{c0, c0, c0, c0};
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) {
                    if constexpr (ABL & 2) sn[t][jq] += __builtin_bit_cast(f32x4, kf[t][s]);
                    else mma_step<T>(sn[t][jq], kf[t][s], qf[jq][s]);
                }
        }
    };
    auto mask_tail = [&](int tile, f32x4(&sn)[4][NJQ]) {  // keys beyond Lk (last tile only)
        const int kv0 = tile * BKV;
        if (kv0 + BKV > Lk) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kv0 + k_row_key(16 * t + 4 * g + r) >= Lk) {
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) sn[t][jq][r] = -INFINITY;
                    }
                }
        }
    };
    auto local_max = [&](const f32x4(&sn)[4][NJQ], float(&mloc)[NJQ]) {
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) {
            float mx = sn[0][jq][0];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sn[t][jq][r]);
            asm volatile("" : "+v"(mx));
            mloc[jq] = mx;
        }
    };
    auto move_reference = [&](const float(&mloc)[NJQ], f32x4(&sn)[4][NJQ], bool first) {  // between iterations: every P V issued so far is in o / lacc
        bool need = first;
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) need |= FOLD ? mloc[jq] > 8.0f : mloc[jq] > mrun[jq] + p.thr;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) {
                const float mx = group_max<true>(mloc[jq]);
                if constexpr (FOLD) {
                    const float d = first ? mx : fmaxf(mx, 0.f);  // the scores of `sn` were taken against the old reference: d is the reference's step
                    const float alpha = fast_exp2(-d);
                    mrun[jq] += d;
                    lacc[jq] *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i][jq] *= alpha;
#pragma unroll
                    for (int t = 0; t < 4; ++t) sn[t][jq] -= d;
                } else {
                    const float mnew = fmaxf(mrun[jq], mx);
                    const float alpha = fast_exp2((mrun[jq] - mnew) * p.c);
                    mrun[jq] = mnew;
                    lacc[jq] *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i][jq] *= alpha;
                }
            }
        }
    };
    auto expo = [&](f32x4(&sc)[4][NJQ], int jq, frag_t(&pb)[2]) {
        const float mc = mrun[jq] * p.c;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (ABL & 1) {
                    pk[r] = (bf16_t)sc[2 * s2][jq][r];
                    pk[4 + r] = (bf16_t)sc[2 * s2 + 1][jq][r];
                } else if constexpr (FOLD) {
                    pk[r] = (bf16_t)fast_exp2(sc[2 * s2][jq][r]);
                    pk[4 + r] = (bf16_t)fast_exp2(sc[2 * s2 + 1][jq][r]);
                } else {
                    pk[r] = (bf16_t)fast_exp2(sc[2 * s2][jq][r] * p.c - mc);
                    pk[4 + r] = (bf16_t)fast_exp2(sc[2 * s2 + 1][jq][r] * p.c - mc);
                }
            }
            pb[s2] = __builtin_bit_cast(frag_t, pk);
            asm volatile("" : "+v"(pb[s2]));  // the packed fragment exists HERE (IR-level sinking otherwise moves the exponentials down to the P V product that uses them)
        }
    };
    auto read_v = [&](const char* vs, frag_t(&vf)[2][4]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 4; ++i) vf[s2][i] = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * s2 + g));
    };
    auto pv = [&](const frag_t(&vf)[2][4], int jq, const frag_t(&pb)[2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if constexpr (ABL & 4) {
                lacc[jq] += __builtin_bit_cast(f32x4, pb[s2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i][jq] += __builtin_bit_cast(f32x4, vf[s2][i]);
                continue;
            }
            mma_step<T>(lacc[jq], ones, pb[s2]);
#pragma unroll
            for (int i = 0; i < 4; ++i) mma_step<T>(o[i][jq], vf[s2][i], pb[s2]);
        }
    };
    // order of one phase: NDS LDS reads and HEADV vector instructions up front, then NM x { one MFMA, PERGAP vector instructions }
    auto pin = [&](auto nds, auto headv, auto nm, auto pergap) {
        if constexpr (SCHED != 0) {
            if constexpr (decltype(nds)::value > 0) __builtin_amdgcn_sched_group_barrier(0x100, decltype(nds)::value, 0);
            if constexpr (decltype(headv)::value > 0) __builtin_amdgcn_sched_group_barrier(0x402, decltype(headv)::value, 0);
#pragma unroll
            for (int m = 0; m < decltype(nm)::value; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (decltype(pergap)::value > 0) __builtin_amdgcn_sched_group_barrier(0x402, decltype(pergap)::value, 0);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;

    // ---- prologue: K(0), V(0), K(1) ----
    if constexpr (DMA) {
        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(ntile > 1 ? 1 : 0, 1);
        wait_vm0();
    } else {
        load_k(0);
        load_v(0);
        commit_k(0);
        commit_v(0);
        load_k(ntile > 1 ? 1 : 0);
        commit_k(1);
    }
    __syncthreads();
    f32x4 sa[4][NJQ], sb[4][NJQ];
    qk(kbuf, sa);
#ifndef MI355X_ATTN_NO_PROLOGUE_BARRIER  // (probing build: what this barrier costs)
    __syncthreads();  // iteration 0 re-fills K buffer 0: every wave's fragment reads of tile 0 come first (inside the loop the end-of-iteration barrier orders them)
#endif
    mask_tail(0, sa);
    {
        float mloc[NJQ];
        local_max(sa, mloc);
        move_reference(mloc, sa, true);
    }

    // one iteration: `sc` = scores of tile i (in), `sn` = scores of tile i + 1 (out, NEXT only)
    auto body = [&](int i, f32x4(&sc)[4][NJQ], f32x4(&sn)[4][NJQ], auto next) {
        constexpr bool NEXT = decltype(next)::value;
        if constexpr (NEXT && !(ABL & 8)) {
            if constexpr (DMA) {
                dma_k(i + 2 < ntile ? i + 2 : ntile - 1, i & 1);
                dma_v(i + 1, (i + 1) & 1);
            } else {
                load_k(i + 2 < ntile ? i + 2 : ntile - 1);
                load_v(i + 1);
            }
        }
        frag_t pb[NJQ][2];
        frag_t vf[2][4];
        // phase 1
        if constexpr (NEXT) qk(kbuf + ((i + 1) & 1) * TILEB, sn);
        expo(sc, 0, pb[0]);
        if constexpr (NEXT) pin(I8{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 8 * NJQ>{}, std::integral_constant<int, (NJQ == 2 ? 2 : 4)>{});
        __builtin_amdgcn_sched_barrier(0);
        read_v(vbuf + (i & 1) * TILEB, vf);
        // middle phases
#pragma unroll
        for (int jq = 1; jq < NJQ; ++jq) {
            pv(vf, jq - 1, pb[jq - 1]);
            expo(sc, jq, pb[jq]);
            pin(I8{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        // last phase
        if constexpr (NEXT) mask_tail(i + 1, sn);  // (a wave-uniform branch: in front of the phase, not inside it)
        pv(vf, NJQ - 1, pb[NJQ - 1]);
        float mloc[NJQ];
        if constexpr (NEXT && (ABL & 32)) {
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) mloc[jq] = sn[0][jq][0];
        }
        if constexpr (NEXT && !(ABL & 32)) {
            local_max(sn, mloc);
            pin(std::integral_constant<int, (NJQ == 1 ? 8 : 0)>{}, I0{}, std::integral_constant<int, 10>{}, std::integral_constant<int, (NJQ == 2 ? 3 : 2)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) {
            move_reference(mloc, sn, false);
            if constexpr (!(ABL & 8)) {
                if constexpr (DMA) {
                    wait_vm0();
                } else {
                    commit_k(i & 1);
                    commit_v((i + 1) & 1);
                }
            }
            if constexpr (!(ABL & 16)) __syncthreads();
        }
    };
    {
        int i = 0;
        for (; i + 2 < ntile; i += 2) {
            body(i, sa, sb, std::true_type{});
            body(i + 1, sb, sa, std::true_type{});
        }
        if (i + 1 < ntile) {  // two tiles left
            body(i, sa, sb, std::true_type{});
            body(i + 1, sb, sa, std::false_type{});
        } else {
            body(i, sa, sb, std::false_type{});
        }
    }

    // ---- store: lane owns d = 16g + 4i + r (16 consecutive) of query 16jq + c16 ----
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        const int qr = q0 + 16 * jq + c16;
        if (qr >= p.Lq) continue;
        const float inv = kv.out_scale / lacc[jq][0];
        T* op = reinterpret_cast<T*>(p.out + (int64_t)b * p.obsb + (int64_t)qr * p.ldob + (int64_t)h * ROWB) + 16 * g;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov.set(e, o[(c * 8 + e) >> 2][jq][e & 3] * inv);
            store16<T>(op + c * 8, ov);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- short K/V
// Cross-attention of the UNet step: 77 text keys (+ 4 image-prompt keys as a second stream) = at most THREE 64-key tiles in total.  The general
// kernel walks them as a chain of load -> barrier -> compute -> barrier per tile and per stream (seven barriers and three dependent memory
// round trips for a few hundred MFMAs); here every tile of every stream is requested up front, lands in its own LDS slot behind ONE barrier,
// and the tiles are then consumed back to back.  Same arithmetic, same order of operations per stream as attn_kernel.
// NJQ = 16-query groups per wave: 2 = 128-query workgroups; 1 = 64-query workgroups (twice the workgroups, half the chain per wave) for grids of few 128-query workgroups
template <typename T, int NSTREAM, int NJQ = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? 2 : 1))) void attn_short_kernel(const AttnP p) {
    constexpr int D = 64, BKV = 64, NW = 4, BQW = 16 * NJQ, SLOTS = 3;
    constexpr int ES = sizeof(T);
    constexpr int ROWB = D * ES;
    constexpr int CPR = ROWB / 16;
    constexpr int NTHR = NW * 64;
    constexpr int TILEB = 64 * ROWB;
    constexpr int STAGE = 2 * TILEB;
    constexpr int LI = 64 * CPR / NTHR;
    constexpr int NS = D / DT<T>::KSTEP;
    constexpr bool IS_BF16 = (ES == 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id();
    const int g = lane >> 4, c16 = lane & 15;
    int bid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qt = bid % p.qtiles;
    bid /= p.qtiles;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int q0 = qt * (BQW * NW) + wid * BQW;

    frag_t qf[NJQ][NS];
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        int qr = q0 + 16 * jq + c16;
        qr = qr < p.Lq ? qr : p.Lq - 1;
        const char* qp = p.q + (int64_t)b * p.qbsb + (int64_t)qr * p.ldqb + (int64_t)h * ROWB;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[jq][s] = *reinterpret_cast<const frag_t*>(qp + (4 * s + g) * 16);
    }

    const int nt0 = (p.kv[0].Lk + BKV - 1) / BKV;
    const int nt1 = NSTREAM > 1 ? (p.kv[1].Lk + BKV - 1) / BKV : 0;
    const int ntot = nt0 + nt1;  // <= SLOTS (checked by the host)

    // ---- every tile of every stream: global -> registers -> its own LDS slot, one barrier ----
    {
        frag_t kr[SLOTS][LI], vr[SLOTS][LI];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (sl < ntot) {
                const int sidx = (NSTREAM > 1 && sl >= nt0) ? 1 : 0;
                const KvP& kv = p.kv[sidx];
                const int kv0 = (sidx ? sl - nt0 : sl) * BKV;
                const char* kbase = kv.k + (int64_t)b * kv.kbsb + (int64_t)h * ROWB;
                const char* vbase = kv.vt + (int64_t)h * D * kv.ldvtb + (int64_t)b * kv.vtbsb;
#pragma unroll
                for (int it = 0; it < LI; ++it) {
                    const int q = it * NTHR + tid, row = q / CPR, pch = q % CPR;
                    const int coff = (pch ^ swz<ROWB>(row)) << 4;
                    int kr_ = kv0 + (IS_BF16 ? k_row_key(row) : row);  // see k_row_key
                    kr_ = kr_ < kv.Lk ? kr_ : kv.Lk - 1;
                    kr[sl][it] = *reinterpret_cast<const frag_t*>(kbase + (int64_t)kr_ * kv.ldkb + coff);
                    const int j = row >> 4, a = (row >> 2) & 3, bb = row & 3;
                    const int vrow = 16 * a + 4 * j + bb;  // head-dim index stored in LDS row `row` of the V^T tile (see attn_kernel)
                    vr[sl][it] = *reinterpret_cast<const frag_t*>(vbase + (int64_t)vrow * kv.ldvtb + (int64_t)kv0 * ES + coff);
                }
            }
        }
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (sl < ntot) {
                char* ks = smem + sl * STAGE;
#pragma unroll
                for (int it = 0; it < LI; ++it) {
                    *reinterpret_cast<frag_t*>(ks + (it * NTHR + tid) * 16) = kr[sl][it];
                    *reinterpret_cast<frag_t*>(ks + TILEB + (it * NTHR + tid) * 16) = vr[sl][it];
                }
            }
        }
    }
    __syncthreads();

    f32x4 res[4][NJQ], o[4][NJQ];
    float mrun[NJQ], lsum[NJQ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) res[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f}, o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) mrun[jq] = -INFINITY, lsum[jq] = 0.f;

    auto finish = [&](float out_scale) {
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) {
            const float l = group_sum<true>(lsum[jq]);
            const float inv = out_scale / l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                res[i][jq] += o[i][jq] * inv;
                o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            mrun[jq] = -INFINITY, lsum[jq] = 0.f;
        }
    };

#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        if (sl < ntot) {
            const int sidx = (NSTREAM > 1 && sl >= nt0) ? 1 : 0;
            if (NSTREAM > 1 && sl == nt0) finish(p.kv[0].out_scale);  // first tile of the second stream: close the first
            const int Lk = p.kv[sidx].Lk;
            const int kv0 = (sidx ? sl - nt0 : sl) * BKV;
            const char* ks = smem + sl * STAGE;
            const char* vs = ks + TILEB;
            // ---- S^T = K Q^T ----
            f32x4 st[4][NJQ];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) st[t][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const frag_t kf = lds_read_frag(ks, tile_off<ROWB>(16 * t + c16, 4 * s + g));
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(st[t][jq], kf, qf[jq][s]);
                }
            }
            if (kv0 + BKV > Lk) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int krow = 16 * t + 4 * g + r;
                        if (kv0 + (IS_BF16 ? k_row_key(krow) : krow) >= Lk) {
#pragma unroll
                            for (int jq = 0; jq < NJQ; ++jq) st[t][jq][r] = -INFINITY;
                        }
                    }
            }
            // ---- online softmax ----
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) {
                float mx = st[0][jq][0];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][jq][r]);
                mx = group_max<true>(mx);
                const float mnew = fmaxf(mrun[jq], mx);
                const float alpha = fast_exp2((mrun[jq] - mnew) * p.c);
                const float mc = mnew * p.c;
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
                for (int i = 0; i < 4; ++i) o[i][jq] *= alpha;
            }
            // ---- O^T += V^T P^T ----
            if constexpr (IS_BF16) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    frag_t pb[NJQ];
#pragma unroll
                    for (int jq = 0; jq < NJQ; ++jq) {
                        bf16x8 pk;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pk[r] = (bf16_t)st[2 * s2][jq][r];
                            pk[4 + r] = (bf16_t)st[2 * s2 + 1][jq][r];
                        }
                        pb[jq] = __builtin_bit_cast(frag_t, pk);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const frag_t vf = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * s2 + g));
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(o[i][jq], vf, pb[jq]);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const frag_t vf = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * t + g));
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(o[i][jq], vf, __builtin_bit_cast(frag_t, st[t][jq]));
                    }
                }
            }
        }
    }
    finish(p.kv[NSTREAM - 1].out_scale);

    // ---- store: lane owns d = 16g + 4i + r (16 consecutive) of query 16jq + c16 ----
    constexpr int EPC = DT<T>::EPC;
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        const int qr = q0 + 16 * jq + c16;
        if (qr >= p.Lq) continue;
        T* op = reinterpret_cast<T*>(p.out + (int64_t)b * p.obsb + (int64_t)qr * p.ldob + (int64_t)h * ROWB) + 16 * g;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * i + r] = res[i][jq][r];
#pragma unroll
        for (int c = 0; c < 16 / EPC; ++c) {
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < EPC; ++e) ov.set(e, v[c * EPC + e]);
            store16<T>(op + c * EPC, ov);
        }
    }
}

// Round 6: the bf16 form of the short-K/V kernel.  Same arithmetic as attn_short_kernel (results agree to the last bf16 digit or one next to it: the compiler pairs
// the row sum's additions differently when a half tile has 8 scores per lane instead of 16), but
//   * every K / V^T tile goes global -> LDS by LDS-DMA (buffer_load ... lds; no staging registers, no ds_write): the kernel was moving 48 KB per workgroup through
//     registers for 10 KB of live keys;
//   * lanes whose keys lie beyond Lk carry the out-of-range offset, for which the hardware writes ZEROS (K rows, whole 8-key chunks of V^T): nothing beyond Lk is
//     read from memory, and nothing depends on what the V^T padding holds except inside the last partly valid chunk;
//   * a tile with at most 32 valid keys (the text stream's second tile: 13 of 64; the image-prompt stream: 4 of 64) is a HALF tile: K rows 32 .. 63 are not
//     requested, Q K^T runs on two 16-key blocks, the softmax on 8 scores per lane and P V on one 32-key step.
template <int NSTREAM, int NJQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_short_dma_kernel(const AttnP p) {
    using T = bf16_t;
    constexpr int D = 64, BKV = 64, NW = 4, BQW = 16 * NJQ, SLOTS = 3, ES = 2, ROWB = D * ES, CPR = ROWB / 16, NTHR = NW * 64, TILEB = 64 * ROWB, STAGE = 2 * TILEB;
    constexpr int LI = 64 * CPR / NTHR, NS = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id();
    const int g = lane >> 4, c16 = lane & 15;
    int bid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int qt = bid % p.qtiles;
    bid /= p.qtiles;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int q0 = qt * (BQW * NW) + wid * BQW;

    const int nt0 = (p.kv[0].Lk + BKV - 1) / BKV;
    const int nt1 = NSTREAM > 1 ? (p.kv[1].Lk + BKV - 1) / BKV : 0;
    const int ntot = nt0 + nt1;  // <= SLOTS (checked by the host)

    // ---- every tile of every stream: LDS-DMA into its own slot ----
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        if (sl < ntot) {
            const int sidx = (NSTREAM > 1 && sl >= nt0) ? 1 : 0;
            const KvP& kv = p.kv[sidx];
            const int kv0 = (sidx ? sl - nt0 : sl) * BKV;
            const bool half = kv.Lk - kv0 <= 32;
            const rsrc_t krs = make_rsrc(kv.k + (int64_t)b * kv.kbsb + (int64_t)h * ROWB + (int64_t)kv0 * kv.ldkb, (int64_t)63 * kv.ldkb + ROWB);
            const rsrc_t vrs = make_rsrc(kv.vt + (int64_t)h * D * kv.ldvtb + (int64_t)b * kv.vtbsb + (int64_t)kv0 * ES, (int64_t)(D - 1) * kv.ldvtb + BKV * ES);
            char* ks = smem + sl * STAGE;
#pragma unroll
            for (int it = 0; it < LI; ++it) {
                const int q = it * NTHR + tid, row = q / CPR, pch = q % CPR;
                const int cl = pch ^ swz<ROWB>(row);  // the logical chunk this lane's physical slot holds
                const int key = k_row_key(row);       // LDS rows 0 .. 31 hold keys 0 .. 31 of the tile
                if (!(half && it * NTHR >= 32 * CPR)) {  // wave-uniform: a half tile has no second K iteration
                    const uint32_t ko = kv0 + key < kv.Lk ? (uint32_t)(key * (int)kv.ldkb + cl * 16) : 0x80000000u;
                    blds16(krs, ks + (it * NTHR + wid * 64) * 16, ko, 0);
                }
                const int j = row >> 4, a = (row >> 2) & 3, bb = row & 3;
                const int vrow = 16 * a + 4 * j + bb;
                const uint32_t vo = kv0 + cl * 8 < kv.Lk ? (uint32_t)(vrow * (int)kv.ldvtb + cl * 16) : 0x80000000u;
                blds16(vrs, ks + TILEB + (it * NTHR + wid * 64) * 16, vo, 0);
            }
        }
    }

    frag_t qf[NJQ][NS];
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        int qr = q0 + 16 * jq + c16;
        qr = qr < p.Lq ? qr : p.Lq - 1;
        const char* qp = p.q + (int64_t)b * p.qbsb + (int64_t)qr * p.ldqb + (int64_t)h * ROWB;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[jq][s] = *reinterpret_cast<const frag_t*>(qp + (4 * s + g) * 16);
    }
    wait_vm0();
    __syncthreads();

    f32x4 res[4][NJQ], o[4][NJQ];
    float mrun[NJQ], lsum[NJQ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) res[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f}, o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) mrun[jq] = -INFINITY, lsum[jq] = 0.f;

    auto finish = [&](float out_scale) {
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) {
            const float l = group_sum<true>(lsum[jq]);
            const float inv = out_scale / l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                res[i][jq] += o[i][jq] * inv;
                o[i][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            mrun[jq] = -INFINITY, lsum[jq] = 0.f;
        }
    };

    auto tile = [&](const char* ks, int kv0, int Lk, auto halfc) {
        constexpr int TT = decltype(halfc)::value ? 2 : 4;  // 16-key blocks of the tile that hold valid keys
        const char* vs = ks + TILEB;
        f32x4 st[TT][NJQ];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) st[t][jq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const frag_t kf = lds_read_frag(ks, tile_off<ROWB>(16 * t + c16, 4 * s + g));
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(st[t][jq], kf, qf[jq][s]);
            }
        }
        if (kv0 + BKV > Lk) {
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kv0 + k_row_key(16 * t + 4 * g + r) >= Lk) {
#pragma unroll
                        for (int jq = 0; jq < NJQ; ++jq) st[t][jq][r] = -INFINITY;
                    }
                }
        }
#pragma unroll
        for (int jq = 0; jq < NJQ; ++jq) {
            float mx = st[0][jq][0];
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][jq][r]);
            mx = group_max<true>(mx);
            const float mnew = fmaxf(mrun[jq], mx);
            const float alpha = fast_exp2((mrun[jq] - mnew) * p.c);
            const float mc = mnew * p.c;
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(st[t][jq][r] * p.c - mc);
                    st[t][jq][r] = e;
                    ps += e;
                }
            lsum[jq] = lsum[jq] * alpha + ps;
            mrun[jq] = mnew;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i][jq] *= alpha;
        }
#pragma unroll
        for (int s2 = 0; s2 < TT / 2; ++s2) {
            frag_t pb[NJQ];
#pragma unroll
            for (int jq = 0; jq < NJQ; ++jq) {
                bf16x8 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pk[r] = (bf16_t)st[2 * s2][jq][r];
                    pk[4 + r] = (bf16_t)st[2 * s2 + 1][jq][r];
                }
                pb[jq] = __builtin_bit_cast(frag_t, pk);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const frag_t vf = lds_read_frag(vs, tile_off<ROWB>(16 * i + c16, 4 * s2 + g));
#pragma unroll
                for (int jq = 0; jq < NJQ; ++jq) mma_step<T>(o[i][jq], vf, pb[jq]);
            }
        }
    };

#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        if (sl < ntot) {
            const int sidx = (NSTREAM > 1 && sl >= nt0) ? 1 : 0;
            if (NSTREAM > 1 && sl == nt0) finish(p.kv[0].out_scale);  // first tile of the second stream: close the first
            const int Lk = p.kv[sidx].Lk;
            const int kv0 = (sidx ? sl - nt0 : sl) * BKV;
            if (Lk - kv0 <= 32) tile(smem + sl * STAGE, kv0, Lk, std::true_type{});
            else tile(smem + sl * STAGE, kv0, Lk, std::false_type{});
        }
    }
    finish(p.kv[NSTREAM - 1].out_scale);

#pragma unroll
    for (int jq = 0; jq < NJQ; ++jq) {
        const int qr = q0 + 16 * jq + c16;
        if (qr >= p.Lq) continue;
        T* op = reinterpret_cast<T*>(p.out + (int64_t)b * p.obsb + (int64_t)qr * p.ldob + (int64_t)h * ROWB) + 16 * g;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov.set(e, res[(c * 8 + e) >> 2][jq][e & 3]);
            store16<T>(op + c * 8, ov);
        }
    }
}

int g_attn_short_q16 = 0;  // short-K/V kernel with 64-query workgroups: 0 = where the grid of 128-query workgroups is short (<= 640), 1 = never, 2 = always

int g_attn_short_dma = 1;  // bf16 launches of the short-K/V kernel take attn_short_dma_kernel (0: the register-staged attn_short_kernel)

template <typename T, int NSTREAM, int NJQ, bool DMA>
auto short_kernel_of() {
    if constexpr (DMA) return attn_short_dma_kernel<NSTREAM, NJQ>;
    else return attn_short_kernel<T, NSTREAM, NJQ>;
}

template <typename T, int NSTREAM, int NJQ = 2, bool DMA = false>
int launch_attn_short(const AttnP& p0, int xcd, hipStream_t stream) {
    if constexpr (sizeof(T) == 2 && !DMA) {
        bool ok = g_attn_short_dma != 0;
        for (int s = 0; s < p0.nstream; ++s) ok = ok && 64 * p0.kv[s].ldkb < 0x7fffffff && 64 * p0.kv[s].ldvtb < 0x7fffffff;  // 32-bit offsets inside a tile
        if (ok) return launch_attn_short<T, NSTREAM, NJQ, true>(p0, xcd, stream);
    }
    constexpr int LDS = 3 * 2 * 64 * 64 * sizeof(T);
    auto kfn = short_kernel_of<T, NSTREAM, NJQ, DMA>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    AttnP p = p0;
    constexpr int BQ = 64 * NJQ;
    p.qtiles = (p.Lq + BQ - 1) / BQ;
    p.xcd = xcd;
    hipLaunchKernelGGL(kfn, dim3(p.qtiles * p.H * p.B), dim3(256), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

int g_attn_glds = 0;   // register-staged K/V loader by default: measured faster than glds for attention (probe_attn3)
int g_attn_depth = 1;  // K/V tiles in flight in the register-staged loader (mi355x_attention_set_pipeline); 2 measured no better (r02_j_probe_attn.log)
int g_attn_xcd = 1;    // q-tiles of a head on one XCD
int g_attn_opt = 13;   // OPT bits of attn_kernel: permlane reductions (+1-4 % on every self-attention shape, r02_k / r02_m probes), lazy running maximum + row sums from the matrix pipe (bf16 one-stream launches: -11 ... -13 %, r06_zi_probe_attn_opt.log)
int g_attn_abl = 0;    // ABL bits (probing)
int g_attn_kvs = 0;    // short grids (see launch_attn_nw): 0 = 16-query waves where the grid is short, 1 = never, 2 = key-split workgroups always, 3 = 16-query waves always (single-stream launches)

template <typename T, int NW, int NSTREAM, bool GLDS, int NJQ = 2, int RD = 2, int OPT = 0, int ABL = 0, int KVS = 1>
int launch_attn(const AttnP& p0, hipStream_t stream) {
    constexpr int LDS = (GLDS ? 3 : 2) * 2 * 64 * 64 * sizeof(T);
    auto kfn = attn_kernel<T, NW, NSTREAM, GLDS, NJQ, RD, OPT, ABL, KVS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    AttnP p = p0;
    constexpr int BQ = 16 * NJQ * NW / KVS;  // queries per workgroup
    p.qtiles = (p.Lq + BQ - 1) / BQ;
    p.xcd = g_attn_xcd;
    const int grid = p.qtiles * p.H * p.B;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

int g_attn_pipe = 3;  // software-pipelined loop (attn_pipe_kernel) for bf16 one-stream launches: 0 = off, 1 = register-staged K/V, 2 = the same without the pinned instruction order (probing), 3 = K/V by LDS-DMA (default: 1024 tokens 26.8 -> 19.2 us, 4096 tokens 133 -> 109 us, r06_zl_probe_attn_dma.log)

int g_attn_pipe_fold = 0;  // attn_pipe_kernel's FOLD (LDS-DMA instances)

template <int NJQ, int SCHED, int ABL = 0, bool DMA = false, bool FOLD = false>
int launch_attn_pipe(const AttnP& p0, hipStream_t stream) {
    constexpr int LDS = 4 * 64 * 64 * 2;
    auto kfn = attn_pipe_kernel<NJQ, SCHED, ABL, DMA, FOLD>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    AttnP p = p0;
    constexpr int BQ = 16 * NJQ * 4;
    p.qtiles = (p.Lq + BQ - 1) / BQ;
    p.xcd = g_attn_xcd;
    hipLaunchKernelGGL(kfn, dim3(p.qtiles * p.H * p.B), dim3(256), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

int g_attn_nw = 0;  // 0 = heuristic, 2 / 4 = force the number of waves (32 queries each) per workgroup

template <typename T, int NW, int OPT>
int launch_attn_opt(const AttnP& p, hipStream_t stream) {
    if constexpr (NW != 2) {  // the 2-wave workgroup stages twice the registers per lane: one set only
        if (g_attn_depth == 2) {
            if constexpr (OPT != 2) {  // (two streams, two register sets, hoisted fragment reads) does not fit 256 registers
                if (p.nstream == 2) return launch_attn<T, NW, 2, false, 2, 2, OPT>(p, stream);
            }
            if (p.nstream == 1) return launch_attn<T, NW, 1, false, 2, 2, OPT>(p, stream);
        }
    }
    if (p.nstream == 2) return launch_attn<T, NW, 2, false, 2, 1, OPT>(p, stream);
    return launch_attn<T, NW, 1, false, 2, 1, OPT>(p, stream);
}

template <typename T, int NW>
int launch_attn_nw(const AttnP& p, hipStream_t stream) {
    if constexpr (NW == 4 && sizeof(T) == 2) {
#ifdef MI355X_ATTN_PIPE_ABL  // probing build (python -m refiners_amd.build_native --variant pipeabl MI355X_ATTN_PIPE_ABL=1)
        if (g_attn_pipe && p.nstream == 1 && g_attn_abl != 0) {
            switch (g_attn_abl) {
                case 1: return launch_attn_pipe<2, 1, 1>(p, stream);
                case 2: return launch_attn_pipe<2, 1, 2>(p, stream);
                case 4: return launch_attn_pipe<2, 1, 4>(p, stream);
                case 6: return launch_attn_pipe<2, 1, 6>(p, stream);
                case 8: return launch_attn_pipe<2, 1, 8>(p, stream);
                case 16: return launch_attn_pipe<2, 1, 16>(p, stream);
                case 24: return launch_attn_pipe<2, 1, 24>(p, stream);
                case 32: return launch_attn_pipe<2, 1, 32>(p, stream);
                case 33: return launch_attn_pipe<2, 1, 33>(p, stream);
                case 39: return launch_attn_pipe<2, 1, 39>(p, stream);
                case 63: return launch_attn_pipe<2, 1, 63>(p, stream);
                case 128: return launch_attn_pipe<2, 1, 0, true>(p, stream);
                case 129: return launch_attn_pipe<2, 1, 1, true>(p, stream);
                case 134: return launch_attn_pipe<2, 1, 6, true>(p, stream);
                case 144: return launch_attn_pipe<2, 1, 16, true>(p, stream);
                case 167: return launch_attn_pipe<2, 1, 39, true>(p, stream);
                case 136: return launch_attn_pipe<2, 1, 8, true>(p, stream);
                case 152: return launch_attn_pipe<2, 1, 24, true>(p, stream);
                case 175: return launch_attn_pipe<2, 1, 47, true>(p, stream);
                case 191: return launch_attn_pipe<2, 1, 63, true>(p, stream);
                default: return MI355X_EARG;
            }
        }
#endif
    }
    if (g_attn_glds) {
        if (p.nstream == 2) return launch_attn<T, NW, 2, true>(p, stream);
        return launch_attn<T, NW, 1, true>(p, stream);
    }
    if constexpr (NW == 4 && sizeof(T) == 2) {
        if (g_attn_abl && p.nstream == 1) {  // probing: where a tile's time goes (wrong results by construction)
            switch (g_attn_abl) {
                case 1: return launch_attn<T, 4, 1, false, 2, 1, 0, 1>(p, stream);
                case 2: return launch_attn<T, 4, 1, false, 2, 1, 0, 2>(p, stream);
                case 4: return launch_attn<T, 4, 1, false, 2, 1, 0, 4>(p, stream);
                case 8: return launch_attn<T, 4, 1, false, 2, 1, 0, 8>(p, stream);
                case 6: return launch_attn<T, 4, 1, false, 2, 1, 0, 6>(p, stream);
                case 14: return launch_attn<T, 4, 1, false, 2, 1, 0, 14>(p, stream);
                case 15: return launch_attn<T, 4, 1, false, 2, 1, 0, 15>(p, stream);
                case 31: return launch_attn<T, 4, 1, false, 2, 1, 0, 31>(p, stream);
                case 63: return launch_attn<T, 4, 1, false, 2, 1, 0, 63>(p, stream);
                case 127: return launch_attn<T, 4, 1, false, 2, 1, 0, 127>(p, stream);
                case 16: return launch_attn<T, 4, 1, false, 2, 1, 0, 16>(p, stream);
                default: return MI355X_EARG;
            }
        }
    }
    if constexpr (NW == 4 && sizeof(T) == 2) {
        if (g_attn_pipe && p.nstream == 1 && g_attn_abl == 0 && g_attn_kvs != 2) {
            const int64_t wg128 = (int64_t)((p.Lq + 127) / 128) * p.H * p.B;
            const bool q16 = g_attn_kvs == 3 || (g_attn_kvs == 0 && wg128 <= 640 && p.kv[0].Lk >= 256);  // (<= 384 for attn_kernel; this loop's 16-query waves win by 2 % at 640 as well)
            if (g_attn_pipe == 2) return q16 ? launch_attn_pipe<1, 0>(p, stream) : launch_attn_pipe<2, 0>(p, stream);
            if (g_attn_pipe == 3) {
                const bool dma_ok = (int64_t)p.kv[0].Lk * p.kv[0].ldkb < 0x7fffffff && 64 * p.kv[0].ldvtb < 0x7fffffff;  // 32-bit offsets inside one (batch, head) slice
                if (dma_ok && g_attn_pipe_fold) return q16 ? launch_attn_pipe<1, 1, 0, true, true>(p, stream) : launch_attn_pipe<2, 1, 0, true, true>(p, stream);
                if (dma_ok) return q16 ? launch_attn_pipe<1, 1, 0, true>(p, stream) : launch_attn_pipe<2, 1, 0, true>(p, stream);
                // (otherwise the register-staged loader below)
            }
            return q16 ? launch_attn_pipe<1, 1>(p, stream) : launch_attn_pipe<2, 1>(p, stream);
        }
    }
    if constexpr (NW == 4) {
        // a grid of 128-query workgroups that leaves most SIMDs with a single wave (a CFG pair's 1024-token self-attention: 320 workgroups):
        // 64-query key-split workgroups instead -- twice the waves, each with half the dependent chain per tile
        const int64_t wg128 = (int64_t)((p.Lq + 127) / 128) * p.H * p.B;
        const bool short_grid = wg128 <= 384 && p.kv[0].Lk >= 256;
        // round 6: such grids run 64-query workgroups of four 16-QUERY waves (twice the waves, each with half the softmax / MFMA chain per tile AND per-wave state small
        // enough for 5-7 resident workgroups): in the step 24.40 -> 24.31 ms against the key-split workgroups of rounds 3-5, which stay available (mode 2)
        // and beat the plain 128-query workgroups (24.48) -- profiles/r06_s_ab_attn_q16.log
        if (p.nstream == 1 && (g_attn_kvs == 3 || (g_attn_kvs == 0 && short_grid))) {
            if constexpr (sizeof(T) == 2) {
                switch (g_attn_opt) {
                    case 5: return launch_attn<T, 4, 1, false, 1, 1, 5>(p, stream);
                    case 9: return launch_attn<T, 4, 1, false, 1, 1, 9>(p, stream);
                    case 13: return launch_attn<T, 4, 1, false, 1, 1, 13>(p, stream);
                    default: break;
                }
            }
            return launch_attn<T, 4, 1, false, 1, 1, 1>(p, stream);
        }
        if (p.nstream == 1 && g_attn_kvs == 2) return (g_attn_opt & 1) ? launch_attn<T, 4, 1, false, 2, 1, 1, 0, 2>(p, stream) : launch_attn<T, 4, 1, false, 2, 1, 0, 0, 2>(p, stream);
    }
    if constexpr (NW == 4 && sizeof(T) == 2) {  // bf16 only: lazy running maximum (bit 2), row sums from the matrix pipe (bit 3)
        if (p.nstream == 1) {
            switch (g_attn_opt) {
                case 5: return launch_attn<T, 4, 1, false, 2, 1, 5>(p, stream);
                case 9: return launch_attn<T, 4, 1, false, 2, 1, 9>(p, stream);
                case 13: return launch_attn<T, 4, 1, false, 2, 1, 13>(p, stream);
                default: break;
            }
        }
    }
    switch (g_attn_opt & 3) {
        case 1: return launch_attn_opt<T, NW, 1>(p, stream);
        case 2: return launch_attn_opt<T, NW, 2>(p, stream);
        case 3: return launch_attn_opt<T, NW, 3>(p, stream);
        default: return launch_attn_opt<T, NW, 0>(p, stream);
    }
}

int g_attn_short = 1;  // all-tiles-up-front kernel for launches of at most three K/V tiles in total (the step's cross-attentions)

template <typename T>
int launch_attn_t(const AttnP& p, hipStream_t stream) {
    if (g_attn_short && !g_attn_glds && g_attn_nw == 0 && g_attn_abl == 0 && g_attn_kvs != 2) {
        int tiles = 0;
        for (int s = 0; s < p.nstream; ++s) tiles += (p.kv[s].Lk + 63) / 64;
        if (tiles <= 3) {
            const int64_t wg128 = (int64_t)((p.Lq + 127) / 128) * p.H * p.B;
            if (g_attn_short_q16 == 2 || (g_attn_short_q16 == 0 && wg128 <= 640))  // (r06_zt_probe_attn_short_dma.log: 2 x 10 x 4096 queries = 640: 12.8 -> 11.6 us; 1280: 21.4 -> 22.3)
                return p.nstream == 2 ? launch_attn_short<T, 2, 1>(p, g_attn_xcd, stream) : launch_attn_short<T, 1, 1>(p, g_attn_xcd, stream);
            return p.nstream == 2 ? launch_attn_short<T, 2>(p, g_attn_xcd, stream) : launch_attn_short<T, 1>(p, g_attn_xcd, stream);
        }
    }
    int nw = g_attn_nw;  // 2 / 4: waves of 32 queries; 14 / 18: 4 / 8 waves of 16 queries
    if (nw == 0) nw = 4;  // 2-wave workgroups never won on MI355X (profiles/r01_c_probe_attention.log)
    if (nw == 14) return p.nstream == 2 ? launch_attn<T, 4, 2, false, 1, 1>(p, stream) : launch_attn<T, 4, 1, false, 1, 1>(p, stream);
    if (nw == 18) return p.nstream == 2 ? launch_attn<T, 8, 2, false, 1, 1>(p, stream) : launch_attn<T, 8, 1, false, 1, 1>(p, stream);
    return nw == 2 ? launch_attn_nw<T, 2>(p, stream) : launch_attn_nw<T, 4>(p, stream);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int mi355x_attention_set_nw(int v) {
    g_attn_nw = v;
    return MI355X_OK;
}

extern "C" int mi355x_attention_set_glds(int v) {
    g_attn_glds = v;
    return MI355X_OK;
}

extern "C" int mi355x_attention_set_pipeline(int tiles_in_flight, int xcd_aware) {  // probing / A-B only, not part of the stable contract
    // tiles_in_flight: bits 0-3 = 1 | 2, bits 4-7 = OPT bits of attn_kernel, bits 8-15 = ABL bits (timing probes, wrong results),
    // bits 16-17 = key-split workgroups: 0 auto, 1 never, 2 always; bit 18 = 1: no short-K/V kernel; bits 19-20 = software-pipelined loop (g_attn_pipe)
    const int d = tiles_in_flight & 15;
    if (d == 1 || d == 2) g_attn_depth = d;
    g_attn_opt = (tiles_in_flight >> 4) & 15;
    g_attn_abl = (tiles_in_flight >> 8) & 255;
    g_attn_kvs = (tiles_in_flight >> 16) & 3;
    g_attn_short = ((tiles_in_flight >> 18) & 1) ? 0 : 1;
    g_attn_pipe = (tiles_in_flight >> 19) & 3;
    g_attn_pipe_fold = (tiles_in_flight >> 21) & 1;
    g_attn_short_q16 = (tiles_in_flight >> 23) & 3;
    g_attn_short_dma = ((tiles_in_flight >> 25) & 1) ? 0 : 1;
    if (xcd_aware >= 0) g_attn_xcd = xcd_aware ? 1 : 0;
    return MI355X_OK;
}

extern "C" int mi355x_attention(const mi355x_attn_args* a, void* stream) {
    if (!a || !a->q || !a->out) return MI355X_EARG;
    if (a->dtype != MI355X_F32 && a->dtype != MI355X_BF16) return MI355X_EDTYPE;
    if (a->D != 64 || a->B <= 0 || a->H <= 0 || a->Lq <= 0 || a->nstream < 1 || a->nstream > 2) return MI355X_ESHAPE;
    const int es = a->dtype == MI355X_F32 ? 4 : 2;
    if (!al16(a->q) || !al16(a->out) || (a->ldq * es) % 16 || (a->ldo * es) % 16 || (a->q_batch_stride * es) % 16 ||
        (a->o_batch_stride * es) % 16)
        return MI355X_ESHAPE;
    AttnP p{};
    p.B = a->B;
    p.H = a->H;
    p.Lq = a->Lq;
    p.nstream = a->nstream;
    p.q = static_cast<const char*>(a->q);
    p.ldqb = a->ldq * es;
    p.qbsb = a->q_batch_stride * es;
    p.out = static_cast<char*>(a->out);
    p.ldob = a->ldo * es;
    p.obsb = a->o_batch_stride * es;
    p.c = a->scale * 1.44269504088896340736f;
    p.thr = p.c > 0.f ? 8.0f / p.c : 0.f;
    for (int s = 0; s < a->nstream; ++s) {
        const mi355x_kv_stream& k = a->kv[s];
        if (!k.k || !k.vt || k.Lk <= 0) return MI355X_ESHAPE;
        if (!al16(k.k) || !al16(k.vt) || (k.ldk * es) % 16 || (k.ldvt * es) % 16 || (k.k_batch_stride * es) % 16 ||
            (k.vt_batch_stride * es) % 16)
            return MI355X_ESHAPE;
        p.kv[s].k = static_cast<const char*>(k.k);
        p.kv[s].vt = static_cast<const char*>(k.vt);
        p.kv[s].ldkb = k.ldk * es;
        p.kv[s].kbsb = k.k_batch_stride * es;
        p.kv[s].ldvtb = k.ldvt * es;
        p.kv[s].vtbsb = k.vt_batch_stride * es;
        p.kv[s].Lk = k.Lk;
        p.kv[s].out_scale = k.out_scale;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->dtype == MI355X_F32) return launch_attn_t<float>(p, st);
    return launch_attn_t<bf16_t>(p, st);
}
