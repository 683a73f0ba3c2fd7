// gemm_params.cuh: launch parameters and small device helpers shared by the GEMM kernels of this directory
// (gemm_kernel.cuh: the 4-wave two-phase loop; gemm8_kernel.cuh: the 8-wave eight-phase loop) and their common epilogue (gemm_epilogue.cuh).
#pragma once
#include <type_traits>

#include "../../include/mi355x_refiners.h"
#include "common.cuh"

namespace mi355x {

struct SegP {
    const char* x;
    const char* w;
    int64_t ldxb;  // bytes
    int64_t ldwb;  // bytes
    int nkb;       // number of 128-byte K blocks in this segment
    int cpb;       // conv: K blocks per tap (= channels*sizeof(T)/128)
    int ksize, stride, ups_shift, H, W;
    int wkb, xkb;  // operand stored K-blocked: [K block][row][128 B]
    int64_t xbytes, wbytes;  // extent of the operand in bytes from x / w (what a buffer descriptor may cover)
    int pad;  // zero rows / columns before the image (ksize / 2, or 0 for the bottom/right-only padding of Downsample(padding=0))
};

struct GemmP {
    int M, N, nseg;
    int OH, OW;
    SegP seg[MI355X_MAX_SEG];
    char* out;
    int64_t ldo;  // elements
    const char* bias;
    const char* rowbias;
    int64_t ld_rowbias;  // elements
    int rows_per_group;
    int geglu;
    int gelu;  // activation on every output column (after bias / row bias, before the residual): 1 = erf-GELU, 2 = x * sigmoid(1.702 x)
    const char* res;
    int64_t ldres;  // elements
    const char* zeros;
    int tiles_m, tiles_n;
    int ksplit, kb_per_split, grid0;  // split-K: ksplit workgroups per tile, each accumulating kb_per_split K blocks
    float* partial;                   // [ksplit][M][N] float32 partial sums (split-K only)
    int tile_hint;                    // 0 = heuristic, 1..6 = caller's choice
    int stage_hint;                   // 0 = heuristic, 2..4 = caller's choice
    int out_kb;                       // GEGLU output stored K-blocked ([column block][M rows][128 B]) for the GEMM that consumes it as x
    // transposed column group: columns n >= nt_begin are stored as out_t[(n - nt_begin) * ldt + m]
    int nt_begin;
    char* out_t;
    int64_t ldt;  // elements
    // LayerNorm folded into this launch (consumer side) / row statistics written by this launch (producer side)
    const float* ln_stats;  // [ln_parts][M][2] (mean, M2) per 32-column chunk of the normalised tensor, or NULL
    int ln_parts;
    float ln_eps;
    const float* ln_s;  // [N]: sum_k W'[n][k]
    const float* ln_c;  // [N]: sum_k beta[k] W[n][k] (+ bias[n])
    float* stats_out;   // [N / 32][M][2], or NULL
    float* colstats;    // GroupNorm statistics of the output (producer side): [ceil(M / 32)][N][2] = per (32-row block, column) (sum, sum of squares)
                        // of the values AS STORED, rows beyond M excluded; or NULL
    int out_f32;        // store `out` as float32 (scores for mi355x_softmax_rows)
    // LoRA inside the launch: per column group g (columns >= lora_nb[g]) the K-blocked stacked down rows, [K blocks][lora_r][128 B];
    // lora_b: [N][lora_r] pre-scaled up-projections (row n = output column n), row-major
    const char* lora_a[3];
    int lora_nb[3];
    int lora_groups;
    int lora_r;            // stacked rank: 32, 64 or 128
    const char* lora_b;
    const float* lora_ls;  // LayerNorm folded into this launch AND LoRA: [groups][lora_r] sum_k A'[r][k] and
    const float* lora_lc;  //                                            [groups][lora_r] sum_k beta[k] A[r][k]
    char* lora_t;          // [groups][M][lora_r] of T: the producers' t = x A^T (already divided by rstd when LayerNorm is folded in)
    int* lora_flags;       // [groups][ceil(M / 32)]: == *lora_epoch once those 32 rows of t are complete
    const int* lora_epoch;
    int lp_blocks;         // producer workgroups at the head of the grid (ceil(M / 32) * groups rounded up to a multiple of 8)
    int lora_tt;           // 1 = no producers: the lp_blocks workgroups at the head of the grid are t-TILES (one per row tile; see gemm_kernel)
    int64_t lora_gs;       // bytes from one group's t to the next: M * lora_r * sizeof(T) rounded up to 128 (a 128-byte line never holds two groups' rows)
    int lora_dbg;          // probing only (mi355x_set_option "lora_dbg", tools/probe_lora.py; timing, not results): 1 = producers exit at once (valid only
                           // while the flags still hold the epoch), 4 = tiles skip the LoRA term entirely, 8 = in-loop hand-off but no product, 16 = product but no hand-off,
                           // 32 = producers at s_setprio 3, 64 = t from producers everywhere, 128 = t from t-tiles wherever the tile is wide enough (results stay right)
    // weight prefetch for the NEXT launch: the first pf_blocks workgroups of the grid do no tile work, they touch every 64 bytes of
    // [pf_ptr, pf_ptr + pf_bytes) so that those lines sit in the Infinity Cache when the next kernel asks for them
    const char* pf_ptr[MI355X_MAX_PREFETCH];
    int64_t pf_bytes[MI355X_MAX_PREFETCH];
    int pf_blocks, pf_mode;  // pf_mode: 1 = plain loads, 2 = non-temporal loads (L2 evict-first)
    int pn, hm, hn;          // XCD rasterisation: the 8 XCDs own a pm x pn grid of hm x hn-tile regions
    int vec_ok;
    // the 8-phase loop's work decomposition (gemm8_kernel.cuh): sk_nk K tiles per output tile; sk_g > 0: "stream-K", sk_g persistent workgroups share
    // tiles x sk_nk units evenly, partial tiles meet in sk_ws ([sk_cap][256 x 256] float32) behind sk_flags ([sk_cap] = 1 while a deposit waits for
    // its owner, 0 between launches; word sk_cap = error flag); sk_order: tile order along the unit axis (0 row-major, 1 column-major)
    int sk_g, sk_nk, sk_order, sk_cap, sk_mode;  // sk_mode 0: workgroup b takes the whole tiles b, b + sk_g, ...; 1: stream-K
    // two-height launches of the 8-wave loop (tile id 11; gemm8_kernel.cuh mix_coords): tiles [0, mix_nbig) are 192 x 256 and cover rows [0, mix_rb) x column tiles [0, mix_cb);
    // the others are 128 x 256 and cover the rest (rows [0, mix_rb) x column tiles [mix_cb, tiles_n), then rows [mix_rb, M) x every column tile).  0: off
    int mix_nbig, mix_rb, mix_cb;
    float* sk_ws;
    int* sk_flags;
};

// Chan's pairwise update of (count, mean, M2); exact for empty operands.
MI_DEV void stat_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float nt = n + nb;
    if (nt > 0.f) {
        const float d = mb - mean, f = nb / nt;
        mean += d * f;
        m2 += m2b + d * d * n * f;
        n = nt;
    }
}

// Column sums over the 16 lanes of a lane group (lane c16 = one row): a[e], b[e] hold this lane's contribution to column e of RUN; on return
// lane c16 holds the 16-lane totals of column `c16 % RUN` in a[0], b[0].  Recursive halving: each exchange adds the partner's half and keeps
// half of the columns (RUN = 8: one full exchange first, the two 8-lane halves then end with the same totals), 15 / 14 exchanges per
// quantity instead of 64 for an all-reduce, a fixed tree (deterministic).  The exchanges are DPP moves inside the 16-lane row (no LDS round
// trip: __shfl_xor compiles to ds_bpermute): partner = row_mirror (15 - c), row_half_mirror (c ^ 7), quad_perm (c ^ 2), (c ^ 1) -- every
// partner differs from the lane in exactly the bit that decides which half it keeps, and the four steps together reach all 16 lanes.
template <int CTRL> MI_DEV float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int RUN> MI_DEV void colsum16(float (&a)[RUN], float (&b)[RUN], int c16) {
    static_assert(RUN == 8 || RUN == 16, "8 or 16 columns per lane");
    constexpr int ROW_MIRROR = 0x140, ROW_HALF_MIRROR = 0x141, QUAD_XOR2 = 0x4E, QUAD_XOR1 = 0xB1;
    auto step = [&](auto ctrl, int w) __attribute__((always_inline)) {  // w = the lane bit of this exchange = number of columns kept
        constexpr int CTRL = decltype(ctrl)::value;
        const bool up = (c16 & w) != 0;
#pragma unroll
        for (int e = 0; e < RUN / 2; ++e) {
            if (e < w) {
                const float sa = up ? a[e] : a[e + w], sb = up ? b[e] : b[e + w];
                const float ka = up ? a[e + w] : a[e], kb = up ? b[e + w] : b[e];
                a[e] = ka + dpp_f32<CTRL>(sa);
                b[e] = kb + dpp_f32<CTRL>(sb);
            }
        }
    };
    if constexpr (RUN == 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] += dpp_f32<ROW_MIRROR>(a[e]);
            b[e] += dpp_f32<ROW_MIRROR>(b[e]);
        }
    } else {
        step(std::integral_constant<int, ROW_MIRROR>{}, 8);
    }
    step(std::integral_constant<int, ROW_HALF_MIRROR>{}, 4);
    step(std::integral_constant<int, QUAD_XOR2>{}, 2);
    step(std::integral_constant<int, QUAD_XOR1>{}, 1);
}

// 8-byte relaxed agent-scope store: sc1 (write-through) on gfx950, the producer half of the hand-off forms the microarchitecture guide lists
// as valid without fences (write-through payload, vmcnt(0), then the flag)
MI_DEV void st_agent8(void* p, uint64_t v) { __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace mi355x
