// gemm_kernel.cuh: the LDS-tiled MFMA GEMM / implicit-GEMM convolution kernel of mi355x_gemm (gfx950), shared by the two
// translation units that instantiate it (gemm.hip: plain GEMMs + the C entry point, gemm_conv.hip: convolutions).
//
//   out[M,N] = epi( sum_s X_s[M,K_s] . W_s[N,K_s]^T )        (see include/mi355x_refiners.h for the contract)
//
// Structure (one workgroup = KG x WM x WN waves, BM x BN output tile, K consumed in 128-byte blocks per row):
//   * both operands are K-contiguous, so an LDS tile is `rows x 128 B`; the global->LDS copy is
//     global_load_lds_dwordx4 (16 B per lane, lane-linear LDS image) with the bank-conflict XOR swizzle applied to
//     the per-lane SOURCE chunk; NSTAGE LDS stages, one barrier per K block (loads of later blocks overlap MFMA on block k);
//   * MFMA orientation: A operand = weight rows, B operand = activation rows, so a lane ends up holding, for each of
//     its activation rows, 4 consecutive N per 16x16 tile.  The weight tile is loaded with the row permutation
//     R = 16j + 4a + b  <->  n = 4*NT*a + 4j + b, which makes every lane own 4*NT CONSECUTIVE output columns:
//     the epilogue (bias, time-embedding row bias, GEGLU, residual, LayerNorm statistics) is fully 16-byte vectorised;
//   * TRANSPOSED column groups (tiles at n0 >= nt_begin): the operand roles are swapped (A = activation rows, with the
//     same permutation applied to THEM), so a lane owns 4*MT consecutive output ROWS of one column and stores them as
//     16-byte vectors into out_t[n][m]: a self-attention's Q | K | V^T come out of ONE launch;
//   * KG = 2 ("K groups"): 8 waves share one output tile, waves 0-3 take the even K blocks, waves 4-7 the odd ones, each
//     group with its own LDS ring; the two partial tiles are added through LDS in a fixed order.  For launches with fewer
//     tiles than CUs this puts two waves on every SIMD (twice the bytes in flight, each wave's LDS / barrier stalls
//     covered by the other) without the HBM round trip of a split-K across workgroups;
//   * conv mode gathers the activation rows straight from the NHWC image (zero padding comes from a zero page, nearest
//     2x upsampling and stride 2 are address arithmetic), so no im2col buffer, no materialised upsample / concat;
//   * LayerNorm folded in: a launch whose x is LN(r) reads r itself; the weights carry gamma (W' = W . diag(gamma)) and the
//     epilogue applies y = rstd[m] * (acc - mean[m] * s[n]) + c[n] with s = W' 1, c = W beta + bias.  mean / rstd come from
//     per-row (mean, M2) partials over 32-column chunks that the launch PRODUCING r wrote from its epilogue (Chan-merged
//     in a fixed order: deterministic, no atomics);
//   * LoRA inside the parent launch (LORA = true): y = x W^T + b + sum_i s_i (x A_i^T) B_i^T from ONE kernel
//     (fluxion/adapters/lora.py:383-397) without any column tile recomputing the down-projection.  ceil(M / 32) x groups extra
//     workgroups at the head of the grid (the "producers", one per 32 rows and column group) compute t = x A_cat^T with this kernel's
//     own loader (plain rows or conv taps) against the stacked down rows A_cat [R][K] (R = 32 / 64 / 128) through an up-to-8-deep LDS
//     ring, round t to the storage type like the reference's intermediate tensor, store it write-through (sc1) to a scratch
//     [groups][M][R] and publish a flag per 32 rows (= the launch's epoch, read from device memory).  Every output tile runs its K loop
//     exactly like an un-adapted launch; three trips before its end the first lanes of wave 0 load the tile's flags, one trip later
//     every thread requests its 16-byte pieces of the t rows (sc1 loads: they land while the last K block is multiplied), and after
//     the loop t goes through LDS, 32 ranks per step, against the pre-scaled up rows (s B_cat) [N][R], whose first 32 ranks the prologue
//     staged into LDS.  Producers have the lowest workgroup ids of the launch, so they are dispatched before any tile that waits for
//     them (a tile that does not find its flags set spins after the loop, bounded by the wall clock, and raises the launch's error word rather than hang).
//     Per-column-group A so that a merged Q|K|V launch keeps its three LoRA sets; Conv2dLora = the same with the conv loader (down
//     conv of the parent's kernel size / stride, 1x1 up conv).  A launch with SEVERAL column groups whose tile is wide enough (Q|K|V^T
//     on the 128-column tile) gets t from "t-tiles" instead: one ordinary tile per row tile at the head of the grid, run against the
//     stacked down rows of all groups as a virtual column tile, whose epilogue publishes t and the flags (GemmP::lora_tt) -- 16
//     workgroups instead of 192 in front of 480 tiles that fill the chip's 512 resident slots;
//   * bf16 -> v_mfma_f32_16x16x32_bf16, f32 (parity mode) -> v_mfma_f32_16x16x4_f32; identical LDS image.
#pragma once
#include "gemm_epilogue.cuh"
#include "gemm_lora_producer.cuh"

namespace mi355x {

#ifndef MI355X_GEMM_PRIO
#define MI355X_GEMM_PRIO 1  // s_setprio 1 around the K loop's MFMA + LDS-read phases, 0 around its wait / barrier / stage-issue section: the co-resident
                            // workgroup's matrix instructions win the issue arbitration against this one's address arithmetic.  Same-process A/B of the whole
                            // step (tools/ab_step.py, profiles/r04_d_ab_gn_prio.log): 25.379 -> 25.314 ms, three interleaved rounds each within 0.01 ms.
                            // (-DMI355X_GEMM_PRIO=0 through refiners_amd.build_native.build_variant rebuilds the old loop for an A/B.)
#endif
#ifndef MI355X_LORA_TT
#define MI355X_LORA_TT 1  // where t comes from t-tiles instead of producers: 0 = nowhere, 1 = launches with more than one column group (Q|K|V^T), 2 = every
                          // launch whose tile is wide enough for groups x rank columns.  (lora_dbg 64 / 128 force 0 / 2 at run time: tools/probe_lora.py)
#endif
#ifndef MI355X_LORA_TT_HOOKS
#define MI355X_LORA_TT_HOOKS 0  // (1: the tiles of a t-tile launch keep the in-loop flag look of the producer design -- A/B builds only)
#endif
// Waves per SIMD the register allocation must leave room for.  Left to itself (launch bounds only) the compiler spreads the LoRA instantiations
// over VGPRs AND AGPRs (128 x 128: 181 + 128, 128 x 64: 141 + 64, 64 x 64: 117 + 32) and they end up ONE wave per SIMD below the un-adapted
// tile of the same size -- a producer workgroup then owns a whole CU while it waits on its loads.  With the un-adapted tile's occupancy as the
// floor the same code fits into 237 / 168 / 117 VGPRs, no AGPR copies, no scratch (hipcc -Rpass-analysis=kernel-resource-usage).
#ifndef MI355X_LORA64_WAVES
#define MI355X_LORA64_WAVES 4  // waves per SIMD the 64 x 64 LoRA instance leaves room for (the plain instance: 84 registers = 5; at 5 the LoRA instance spills 74 registers and the step loses 1.3 %: 24.15 -> 24.47 ms, profiles/r06_zf_ab_lora64_waves.log)
#endif
template <typename T, int BM, int BN, bool LORA> constexpr int gemm_min_waves() {
    if (!LORA || sizeof(T) != 2) return 1;
    return BM * BN == 128 * 128 ? 2 : BM * BN == 64 * 64 ? MI355X_LORA64_WAVES : BM == 128 ? 3 : 2;
}

template <typename T, int BM, int BN, int WM, int WN, bool CONV, int NSTAGE, int KG = 1, bool LORA = false>
__global__ __launch_bounds__(WM* WN * 64 * KG) __attribute__((amdgpu_waves_per_eu(gemm_min_waves<T, BM, BN, LORA>()))) void gemm_kernel(const GemmP p) {
    constexpr int NW = WM * WN;           // waves per K group
    constexpr int NTHR = NW * 64;         // threads per K group: the loader geometry
    constexpr int NTHR_ALL = NTHR * KG;   // threads per workgroup
    constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
    constexpr int XI = BM * 8 / NTHR, WI = BN * 8 / NTHR;
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 LDS stages");
    static_assert(!LORA || (KG == 1 && NTHR == 256 && WN == 2 && NSTAGE == 2), "in-launch LoRA: 4 waves as 2 x 2, two LDS stages");
    static_assert(KG == 1 || KG == 2, "one or two K groups");
    constexpr int WNE = 16 * NT;  // columns per wave
    constexpr int WME = 16 * MT;  // rows per wave
    static_assert(BM * 8 % NTHR == 0 && BN * 8 % NTHR == 0, "tile/thread mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* rowstat = reinterpret_cast<float*>(smem + KG * NSTAGE * STAGE);  // [BM][2] (mean, rstd) of the tile's rows (LayerNorm consumer)
    char* const lora_b0 = smem + KG * NSTAGE * STAGE + (p.ln_stats ? BM * 8 : 0);  // LORA: [BN][32 ranks] of T, the first up-projection chunk (staged by the prologue)

    const int tid_all = threadIdx.x, wid_all = wave_id();
    const int kg = KG > 1 ? wid_all / NW : 0;
    const int tid = tid_all - kg * NTHR, lane = tid & 63, wid = wid_all - kg * NW;
    const int g = lane >> 4, c16 = lane & 15;
    const int wm = wid / WN, wn = wid % WN;
    char* const smem_g = smem + kg * (NSTAGE * STAGE);
    // XCD-aware rasterisation: workgroup b runs on XCD b % 8 (observed dispatch rule, a speed assumption only); each XCD
    // owns one rectangular region of the tile grid so that its private L2 sees as few distinct operand rows as possible.
    if ((int)blockIdx.x < p.pf_blocks) {  // prefetch role (see GemmP::pf_ptr): one 4-byte read per 64 bytes, 8 independent loads in flight
        int acc = 0;
        const int64_t stride = (int64_t)p.pf_blocks * NTHR_ALL * 64;
        constexpr int U = 8;
#pragma unroll
        for (int sp = 0; sp < MI355X_MAX_PREFETCH; ++sp) {
            const char* base = p.pf_ptr[sp];
            const int64_t bytes = base ? p.pf_bytes[sp] : 0;
            for (int64_t off = ((int64_t)blockIdx.x * NTHR_ALL + tid_all) * 64; off < bytes; off += stride * U) {
                int v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t o = off + u * stride;
                    const int* src = reinterpret_cast<const int*>(base + (o < bytes ? o : off));
                    v[u] = p.pf_mode == 2 ? __builtin_nontemporal_load(src) : *src;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= v[u];
            }
        }
        if (acc == 0x5a5a1234 && p.pf_bytes[0] < 0) *reinterpret_cast<int*>(p.out) = acc;  // never taken: keeps the loads alive
        return;
    }
    int bid = (int)blockIdx.x - p.pf_blocks;
    // LORA, t-tile role (GemmP::lora_tt): workgroup `bid` of the grid's head runs the ORDINARY tile path on row tile `bid` against a virtual column tile
    // whose "weights" are the stacked down rows of all column groups (virtual column v = group v / R, rank v % R; columns past groups x R re-read
    // rank 0 and are dropped), and publishes t + the row blocks' flags instead of an output tile.  Same K loop as its neighbours, so t arrives when
    // they leave theirs -- but a launch whose tiles fill the chip's resident slots exactly (Q|K|V^T at a CFG pair: 480 tiles of 128 x 128 on 2 x 256
    // slots) no longer starts a third of them one producer-duration late behind 192 producer workgroups.
    bool ttile = false;
    int tt_tag = 0;
    if constexpr (LORA && !CONV) {
        if (p.lora_tt && bid < p.lp_blocks) {
            if ((p.lora_dbg & 1) || bid >= p.tiles_m) return;  // (probing) / padding up to a multiple of 8
            ttile = true;
            tt_tag = *p.lora_epoch;  // (requested now: by the epilogue it has long arrived)
        }
    }
    if constexpr (LORA) {
        if (!ttile && bid < p.lp_blocks) {  // LoRA producer role: see lora_producer (a separate function; nothing of it lives in the tiles' path)
            if ((p.lora_dbg & 1) || bid >= (p.M + LORA_PM - 1) / LORA_PM * p.lora_groups) return;  // (probing) / padding up to a multiple of 8
            constexpr int RING = KG * NSTAGE * (BM + BN) * 128;
            constexpr int PB1 = (LORA_PM + 32) * 128, PB2 = (LORA_PM + 64) * 128, PB4 = (LORA_PM + 128) * 128;  // bytes per producer stage
            constexpr int P1 = RING / PB1 < 8 ? RING / PB1 : 8, P2 = RING / PB2 < 8 ? RING / PB2 : 8, P4 = RING / PB4;
            if (p.lora_r == 32) lora_producer<T, CONV, 1, P1>(p, bid);
            else if (p.lora_r == 64) lora_producer<T, CONV, 2, P2>(p, bid);
            else if constexpr (P4 >= 2) lora_producer<T, CONV, 4, (P4 < 8 ? P4 : 8)>(p, bid);  // (the host routes rank-128 launches to the 128-column tiles)
            return;
        }
        if (!ttile) bid -= p.lp_blocks;
    }
    const int split = p.ksplit > 1 ? bid / p.grid0 : 0;
    const int bx = bid - split * p.grid0;
    int tm, tn;
    if (ttile) {
        tm = bid;
        tn = 0;
    } else if (p.pn > 0) {  // rectangular regions (exact split of the tile grid, grid0 = 8 * hm * hn)
        const int xcd = bx & 7, idx = bx >> 3;
        const int rm = xcd / p.pn, rn = xcd - rm * p.pn;
        const int lm = idx / p.hn, ln = idx - lm * p.hn;
        tm = rm * p.hm + lm;
        tn = rn * p.hn + ln;
    } else {  // contiguous chunk of the row-major (pn == 0) or column-major (pn == -1) tile order per XCD, balanced to +-1 tile
        const int id = xcd_remap(bx, p.grid0);
        if (p.pn == 0) {
            tm = id / p.tiles_n;
            tn = id - tm * p.tiles_n;
        } else {
            tn = id / p.tiles_m;
            tm = id - tn * p.tiles_m;
        }
    }
    if (tm >= p.tiles_m || tn >= p.tiles_n) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const bool tr = !CONV && !ttile && n0 >= p.nt_begin;  // workgroup-uniform: this tile is stored transposed (operand roles swapped)

    // ---- per-thread loader coordinates (fixed for the whole K loop) ----
    // Exactly one operand's rows are permuted inside each wave's 16*T-row span (see the header): the weights' normally, the
    // activations' for a transposed tile.
    int xm[XI];      // clamped global row (plain) / global row (conv)
    int xcoff[XI];   // logical chunk * 16
    int xb[XI], xoy[XI], xox[XI];
    bool xvalid[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int q = it * NTHR + tid, row = q >> 3, pch = q & 7;
        xcoff[it] = (pch ^ swz<128>(row)) << 4;
        int mr = row;
        if (!CONV && tr) {
            const int rl = row % WME;
            mr = (row - rl) + 4 * MT * ((rl >> 2) & 3) + 4 * (rl >> 4) + (rl & 3);
        }
        const int m = m0 + mr;
        xvalid[it] = m < p.M;
        xm[it] = m < p.M ? m : p.M - 1;
        if constexpr (CONV) {
            const int ohw = p.OH * p.OW;
            const int b = xm[it] / ohw, rem = xm[it] - b * ohw;
            xb[it] = b;
            xoy[it] = rem / p.OW;
            xox[it] = rem - xoy[it] * p.OW;
        }
    }
    int wnrow[WI], wcoff[WI];
#pragma unroll
    for (int it = 0; it < WI; ++it) {
        const int q = it * NTHR + tid, row = q >> 3, pch = q & 7;
        wcoff[it] = (pch ^ swz<128>(row)) << 4;
        const int rl = row % WNE, j = rl >> 4, a = (rl >> 2) & 3, b = rl & 3;
        const int n = n0 + (tr ? row : (row - rl) + 4 * NT * a + 4 * j + b);
        wnrow[it] = n < p.N ? n : p.N - 1;
        if (LORA && ttile) wnrow[it] = n < p.lora_groups * p.lora_r ? n : 0;  // virtual column (n0 = 0)
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // in-launch LoRA: column group of this tile, and whether this workgroup adds the LoRA term (split-K: the first split only)
    const int lgi = LORA && !ttile ? (p.lora_groups > 1 && n0 >= p.lora_nb[1] ? 1 : 0) + (p.lora_groups > 2 && n0 >= p.lora_nb[2] ? 1 : 0) : 0;
    const bool lora_tail = LORA && !ttile && split == 0 && !(p.lora_dbg & 4);

    // ---- K-block iteration state ----
    int seg = 0, kb = 0;  // kb = block index inside the current segment
    int total_kb = 0;
    for (int s = 0; s < p.nseg; ++s) total_kb += p.seg[s].nkb;
    if (p.ksplit > 1) {  // this workgroup's share of the K blocks: [first, first + total_kb)
        const int first = split * p.kb_per_split;
        total_kb = min(p.kb_per_split, total_kb - first);
        kb = first;
        while (seg < p.nseg - 1 && kb >= p.seg[seg].nkb) {
            kb -= p.seg[seg].nkb;
            ++seg;
        }
    }
    // K groups interleave: group kg takes blocks kg, kg + KG, ... of the workgroup's range
    const int my_kb = (total_kb - kg + KG - 1) / KG;   // blocks of this group
    const int max_kb = (total_kb + KG - 1) / KG;       // loop trips of the workgroup (= group 0's blocks)

    // ---- loader state: everything that does not change from one K block to the next is hoisted out of the loop.
    // Per thread: the row base pointers of the current segment (conv: of the current tap) with the swizzled chunk offset folded
    // in; per workgroup: a running byte offset along K.  The per-iteration cost of a load is one 64-bit add; the segment
    // descriptor (a dynamically indexed kernel argument, i.e. scalar loads + waits) is touched only when the segment or tap changes.
    const char* xbase[XI];
    const char* wbase[WI];
    int64_t xstep = 128, wstep = 128;  // bytes from one K block to the next (row-major: 128; K-blocked: rows * 128)
    int64_t xoff = 0, woff = 0;        // running offsets inside the current segment
    int cur_nkb = 0, cur_cpb = 1, tap = 0, cb = 0;
    auto set_tap = [&](const SegP& sp) __attribute__((always_inline)) {  // conv: per-thread pixel pointers of tap `tap` (zero page for padding / out-of-tile rows)
        int dy = tap / sp.ksize, dx = tap - dy * sp.ksize;
        dy -= sp.pad;
        dx -= sp.pad;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int iy = xoy[it] * sp.stride + dy, ix = xox[it] * sp.stride + dx;
            const int HH = sp.H << sp.ups_shift, WW = sp.W << sp.ups_shift;
            const bool ok = xvalid[it] && iy >= 0 && iy < HH && ix >= 0 && ix < WW;
            const int sy = iy >> sp.ups_shift, sx = ix >> sp.ups_shift;
            const int64_t pix = ((int64_t)xb[it] * sp.H + sy) * sp.W + sx;
            xbase[it] = ok ? sp.x + pix * sp.ldxb + xcoff[it] : nullptr;
        }
    };
    auto enter = [&](int s, int kb0) __attribute__((always_inline)) {  // make segment s current, positioned at its K block kb0
        const SegP& sp = p.seg[s];
        cur_nkb = sp.nkb;
        cur_cpb = sp.cpb;
        wstep = sp.wkb ? (int64_t)p.N * 128 : 128;
        woff = (int64_t)kb0 * wstep;
#pragma unroll
        for (int it = 0; it < WI; ++it) wbase[it] = sp.w + (sp.wkb ? (int64_t)wnrow[it] * 128 : (int64_t)wnrow[it] * sp.ldwb) + wcoff[it];
        if constexpr (CONV) {
            tap = kb0 / sp.cpb;
            cb = kb0 - tap * sp.cpb;
            set_tap(sp);
        } else {
            xstep = sp.xkb ? (int64_t)p.M * 128 : 128;
            xoff = (int64_t)kb0 * xstep;
#pragma unroll
            for (int it = 0; it < XI; ++it) xbase[it] = sp.x + (sp.xkb ? (int64_t)xm[it] * 128 : (int64_t)xm[it] * sp.ldxb) + xcoff[it];
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {  // one K block forward; cross into the next tap / segment when this one is exhausted
        if (seg >= p.nseg) return;
        ++kb;
        woff += wstep;
        if constexpr (CONV) {
            if (++cb == cur_cpb) {
                cb = 0;
                ++tap;
                if (kb < cur_nkb) set_tap(p.seg[seg]);
            }
        } else {
            xoff += xstep;
        }
        if (kb == cur_nkb) {
            kb = 0;
            ++seg;
            if (seg < p.nseg) enter(seg, 0);
        }
    };
    enter(seg, kb);
    if constexpr (LORA && !CONV) {
        if (ttile) {  // (one segment, no split-K: enter() is not called again)
            const int rsh = p.lora_r == 32 ? 5 : p.lora_r == 64 ? 6 : 7;
            wstep = (int64_t)p.lora_r * 128;
            woff = 0;
#pragma unroll
            for (int it = 0; it < WI; ++it) {
                const int gi = wnrow[it] >> rsh, r = wnrow[it] & (p.lora_r - 1);
                const char* ab = gi == 0 ? p.lora_a[0] : gi == 1 ? p.lora_a[1] : p.lora_a[2];
                wbase[it] = ab + (int64_t)r * 128 + wcoff[it];
            }
            // (Tried: asking for every line of the down rows up front, through the LDS-DMA path into LDS nobody reads -- they are not covered by the weight
            //  prefetch and come from HBM in a replayed step.  Nothing for these launches: 25.899 vs 25.909 ms, profiles/r04_k_ab_tt.log.)
            if (p.ln_stats && wid == 0) {
                // sA / cA of the epilogue: small, cold (HBM in a replayed step) and on the path between this tile's last MFMA and the flags every tile
                // of the row block is waiting for -- staged now into the up rows' LDS slot (a t-tile stages none), older than every stage load.
                // (All 64 lanes of wave 0, 16 bytes each: 1 KB per vector, of which the first groups x R floats are real; the rest re-read the last piece.)
                const int piece = min(lane, p.lora_groups * p.lora_r / 4 - 1);
                glds16(p.lora_ls + 4 * piece, lora_b0);
                glds16(p.lora_lc + 4 * piece, lora_b0 + 1024);
            }
        }
    }
    if constexpr (KG > 1) {
        if (kg == 1) advance();
    }

    auto issue = [&](int buf) __attribute__((always_inline)) {
        char* xs = smem_g + buf * STAGE;
        char* ws = xs + XBYTES;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const char* src;
            if constexpr (CONV) src = xbase[it] ? xbase[it] + (int64_t)cb * 128 : p.zeros + xcoff[it];
            else src = xbase[it] + xoff;
            glds16(src, xs + (it * NTHR + wid * 64) * 16);
        }
#pragma unroll
        for (int it = 0; it < WI; ++it) glds16(wbase[it] + woff, ws + (it * NTHR + wid * 64) * 16);
#pragma unroll
        for (int a = 0; a < KG; ++a) advance();
    };

    // ---- software pipeline: NSTAGE LDS buffers, D = NSTAGE - 1 K blocks in flight -------------------------------------
    // per iteration: counted vmcnt (block t has landed, the D-1 younger ones stay in flight) -> raw barrier (no vmcnt(0)
    // drain, guide section 5 "pipelining across barriers") -> issue block t+D into the buffer block t-1 was computed from
    // -> MFMA on block t.  One barrier per K block.
    constexpr int D = NSTAGE - 1;
    constexpr int LPS = XI + WI;  // global_load_lds instructions per thread per stage
    if constexpr (LORA) {
        // the first 32 ranks of this tile's pre-scaled up rows go to LDS now (static weights): by the time the K loop is over they have
        // long landed.  Issued BEFORE the stages, so every counted vmcnt wait of the loop covers them.
        if (lora_tail) {
            constexpr int RB = LORA_RC * (int)sizeof(T), CPRB = RB / 16, BI = BN * CPRB / NTHR;
            static_assert(BN * CPRB % NTHR == 0, "up-projection tile / thread mismatch");
#pragma unroll
            for (int it = 0; it < BI; ++it) {
                const int q = it * NTHR + tid, row = q / CPRB, ch = q % CPRB;
                const int rl = row % WNE, j = rl >> 4, a = (rl >> 2) & 3, b = rl & 3;
                int n = n0 + (tr ? row : (row - rl) + 4 * NT * a + 4 * j + b);
                n = n < p.N ? n : p.N - 1;
                glds16(p.lora_b + ((int64_t)n * p.lora_r) * (int)sizeof(T) + ch * 16, lora_b0 + (it * NTHR + wid * 64) * 16);
            }
        }
    }
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (s0 < my_kb) issue(s0);

    if (p.ln_stats) ln_rowstat<BM, NTHR_ALL>(p, m0, tid_all, rowstat);

    // ---- main loop, software-pipelined through REGISTERS as well: the MFMAs of one half K block (32 bf16 / 16 f32 deep) run
    // while the fragments of the next half are being read from LDS, so no LDS latency is exposed to the matrix pipe:
    //   phase A(t): MFMA on F0(t) [first half of block t]   || ds_read F1(t)
    //   -- counted vmcnt: block t+1 has landed; lgkmcnt(0): this wave is done reading block t; barrier; issue block t+1+D --
    //   phase B(t): MFMA on F1(t)                            || ds_read F0(t+1)
    // One barrier per K block, between the phases.  The interleave inside a phase is pinned with sched_group_barrier (the
    // machine scheduler otherwise sinks every ds_read to just before its first use and waits lgkmcnt(0) eight times per block).
    // LORA, output tiles: the hand-off is started INSIDE the K loop so that its two dependent round trips (flags, then the t rows) ride
    // in the loop's own load stream (a dependent load issued from a streaming CU waits 1-3 us in that CU's memory queue: microarchitecture
    // guide, handoff-1to1).  At trip hook_t = max_kb - 4 the first lanes of every wave load the row block's flags (and the launch's epoch)
    // right behind that trip's stage issue; one trip later -- the loop's vmcnt(0) has retired them -- the wave compares them (a ballot:
    // no cross-wave hand-off, no LDS word) and every lane loads its t values STRAIGHT INTO MFMA FRAGMENT LAYOUT together with the loop's last stage
    // (a wave's 16 rows x 64 B of one row block are 1 KB contiguous: the layout a ds_read of an LDS copy would return is the layout in
    // memory, so t needs no LDS staging and the product after the loop no barrier).  Everything the hooks need is computed inside them
    // (behind an opaque zero, so that nothing is hoisted and kept live through the loop).
    //
    // Why the fast path may use PLAIN (L1 / L2 cacheable) loads although the writers are other CUs, possibly on other XCDs: every t
    // line and flag word is written once per launch, BEFORE its flag is published (write-through stores, acknowledged, then the flag),
    // and this CU's L1 / this XCD's L2 hold no copy from an earlier launch (a kernel boundary makes earlier kernels' writes visible to
    // plain loads of later ones -- what every two-kernel HIP program relies on -- so no stale copy survives it).  Within the launch a
    // cached copy of a t line can only have been fetched by a tile that had already seen its flag set, i.e. it holds the final data;
    // a cached flag word can only be stale in the harmless direction ("not set yet"), which sends the tile to the slow path after
    // the loop: L1-bypassing agent-scope atomics (2-3 us per dependent access from a streaming CU, which is why they are not the default).
    constexpr int L_RB = LORA_RC * (int)sizeof(T), L_KS = LORA_RC / DT<T>::KSTEP;  // bytes per 32-rank row / MMA steps per 32 ranks (bf16: 1, f32: 2)
    frag_t tfr[LORA ? MT : 1][L_KS];
    int lora_fl = 0, lora_tg = 1, lora_ok = 0;
    bool lora_mine = false;
    // (t-tile launches: t is published when the tiles leave their loops, an in-loop look at the flags never finds them set -- no hooks, straight to the wait)
    const int hook_t = (lora_tail && max_kb >= 4 && !(p.lora_dbg & 16) && !(p.lora_tt && MI355X_LORA_TT_HOOKS == 0)) ? max_kb - 4 : -2;
    auto opaque0 = [&]() __attribute__((always_inline)) {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        return z;
    };
    auto lora_poll = [&]() __attribute__((always_inline)) {  // lanes 0 .. BM / 32 - 1 of EVERY wave: one flag each (no cross-wave hand-off needed)
        {
            const int z = opaque0();
            const int nfl = (p.M + LORA_PM - 1) / LORA_PM, fb = m0 / LORA_PM + lane + z;
            // both are VECTOR loads (address through z): nothing here waits -- a scalar load of the epoch would park wave 0 on lgkmcnt(0)
            // for a memory round trip, and with it the workgroup's next barrier
            lora_tg = p.lora_epoch[z];
            const bool mine = lane < BM / LORA_PM && fb < nfl;
            lora_fl = p.lora_flags[mine ? lgi * nfl + fb : 0];  // plain load (see above); lanes without a row block compare equal below
            lora_mine = mine;
        }
    };
    auto lora_load_t = [&](int c) __attribute__((always_inline)) {  // this lane's t fragments of rank chunk c: rows 16 i + c16 of the wave's row span
        const int z = opaque0();
        const char* tg = p.lora_t + lgi * p.lora_gs + (int64_t)(c * LORA_RC + z) * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wm * WME + 16 * i + c16;
            int mr = row;
            if (tr) {
                const int rl = row % WME;
                mr = (row - rl) + 4 * MT * ((rl >> 2) & 3) + 4 * (rl >> 4) + (rl & 3);
            }
            const int m = min(m0 + mr, p.M - 1);
#pragma unroll
            for (int ks = 0; ks < L_KS; ++ks) tfr[i][ks] = *reinterpret_cast<const frag_t*>(tg + (int64_t)m * p.lora_r * (int)sizeof(T) + (4 * ks + g) * 16);
        }
    };
    auto mainloop = [&](auto trc) {
        constexpr bool TR = decltype(trc)::value;
        constexpr int RR = MT + NT;                                // ds_read_b128 per phase
        constexpr int MM = MT * NT * (sizeof(T) == 4 ? 4 : 1);      // MFMA instructions per phase
        frag_t xf0[MT], wf0[NT], xf1[MT], wf1[NT];
        auto read_half = [&](frag_t(&xf)[MT], frag_t(&wf)[NT], int blk, int kk) {
            const char* xs = smem_g + (blk % NSTAGE) * STAGE;
            const char* ws = xs + XBYTES;
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = lds_read_frag(xs, tile_off<128>(wm * WME + 16 * i + c16, 4 * kk + g));
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = lds_read_frag(ws, tile_off<128>(wn * WNE + 16 * j + c16, 4 * kk + g));
        };
        auto mma_half = [&](frag_t(&xf)[MT], frag_t(&wf)[NT]) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (TR) mma_step<T>(acc[i][j], xf[i], wf[j]);
                    else mma_step<T>(acc[i][j], wf[j], xf[i]);
                }
            }
        };
        auto pin = [&]() {  // the phase's LDS reads go out early, one per MFMA, so that the last one is >= MM - RR MFMAs old at the phase end
            constexpr int HEAD = RR < MM ? RR : MM;
#pragma unroll
            for (int r = 0; r < HEAD; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA first: the wait in front of it covers only the previous
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // phase's reads, which are at least MM - RR MFMAs old
            }
            if constexpr (RR > HEAD) __builtin_amdgcn_sched_group_barrier(0x100, RR - HEAD, 0);
            if constexpr (MM > HEAD) __builtin_amdgcn_sched_group_barrier(0x008, MM - HEAD, 0);
        };
        // block 0 lands
        if (D <= my_kb) wait_vm<(D - 1) * LPS>();
        else wait_vm0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (D < my_kb) issue(D % NSTAGE);
        if (my_kb > 0) read_half(xf0, wf0, 0, 0);
        // LORA: the loop is PEELED -- trips [0, hook_t) run a copy without any hand-off code (instruction for instruction the un-adapted
        // kernel's loop: even a few wave-uniform compares and branches between a trip's barrier and its stage issue cost every trip
        // ~70 cycles of the barrier -> issue -> landed critical path, 0.7 us per tile: profiles/r03_g_bisect.log), the last four trips the
        // two trips that carry the hand-off run the copy with the hooks, the last two the plain copy again.
        auto trips = [&](auto hookc, int t_begin, int t_end) __attribute__((always_inline)) {
        constexpr bool HOOKS = decltype(hookc)::value;
        for (int t = t_begin; t < t_end; ++t) {
            const bool active = KG == 1 || t < my_kb;  // the odd group of an odd block count idles through the last trip
            if (active) {  // phase A
                if constexpr (MI355X_GEMM_PRIO != 0) __builtin_amdgcn_s_setprio(1);
                read_half(xf1, wf1, t, 1);
                mma_half(xf0, wf0);
                pin();
            }
            if (t + 1 < max_kb) {
                if constexpr (MI355X_GEMM_PRIO != 0) __builtin_amdgcn_s_setprio(0);
                if (t + 1 + D <= my_kb) wait_vm<(D - 1) * LPS>();
                else wait_vm0();

                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t + 1 + D < my_kb) issue((t + 1 + D) % NSTAGE);
                if constexpr (HOOKS) {
                    if (t == hook_t) lora_poll();
                    if (t == hook_t + 1) {  // the flags loaded one trip ago have arrived (D = 1: this trip's vmcnt(0) drained the queue)
                        lora_ok = __builtin_amdgcn_ballot_w64(!lora_mine || lora_fl == lora_tg) == ~0ull ? 1 : 0;  // wave-uniform
                        if (lora_ok) lora_load_t(0);
                    }
                }
            }
            if (active) {  // phase B
                if constexpr (MI355X_GEMM_PRIO != 0) __builtin_amdgcn_s_setprio(1);
                read_half(xf0, wf0, t + 1, 0);  // (past the last block: a harmless read of a stale stage; keeps the phase one basic block)
                mma_half(xf1, wf1);
                pin();
            }
        }
        };
        if constexpr (LORA) {
            const int ts = hook_t >= 0 ? hook_t : max_kb, te = hook_t >= 0 ? hook_t + 2 : max_kb;
            trips(std::false_type{}, 0, ts);
            trips(std::true_type{}, ts, te);   // the two trips that carry the hand-off
            trips(std::false_type{}, te, max_kb);
        } else {
            trips(std::false_type{}, 0, max_kb);
        }
    };
    if constexpr (CONV) {
        mainloop(std::false_type{});
    } else {
        if (tr) mainloop(std::true_type{});
        else mainloop(std::false_type{});
    }

    if constexpr (LORA) {
        // ---- up-projection: acc += T(t) . (s B_cat)^T, 32 ranks per MMA chunk; t comes from this row block's producers ----------------------
        if (lora_tail && !(p.lora_dbg & 8)) {
            constexpr int RB = L_RB, CPRB = RB / 16, BI = BN * CPRB / NTHR;
            const int nch = p.lora_r / LORA_RC;
            if (p.lora_dbg & 16) lora_ok = 1;  // (probing: no hand-off at all, the product runs on whatever the registers hold)
            if (!lora_ok) {  // this wave did not see its flags set from inside the loop (a short K loop, or a producer still running): wait here, bounded by the wall clock
                {
                    const int nfl = (p.M + LORA_PM - 1) / LORA_PM, fb = m0 / LORA_PM + lane;
                    const int tag = *p.lora_epoch;
                    if (lane < BM / LORA_PM && fb < nfl) {
                        const int* fp = p.lora_flags + lgi * nfl + fb;
                        const uint64_t t0 = wall_clock64();
                        while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {  // L1-bypassing: correct whatever this CU has cached
                            __builtin_amdgcn_s_sleep(2);
                            if (wall_clock64() - t0 > 200000000ull) {
                                // 2 s of the 100 MHz clock: a lost producer must neither hang the GPU nor kill the process's HIP context (a trap would): the
                                // launch's error word -- the int32 behind the last flag -- is raised, the tile goes on with whatever t holds, and the host turns
                                // the word into an error (native.LoraSync.check)
                                __hip_atomic_store(p.lora_flags + p.lora_groups * nfl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                }
                lora_load_t(0);
            }
            for (int c = 0; c < nch; ++c) {
                const char* bs = lora_b0;  // chunk 0: staged by the prologue, made visible by the K loop's barriers
                if (c) {  // ranks beyond 32: t straight from memory again, the up rows through the (drained) stage buffers
                    __syncthreads();  // every wave is done with the stage buffers / with the previous chunk's rows
                    lora_load_t(c);
                    char* bl = smem;
#pragma unroll
                    for (int it = 0; it < BI; ++it) {
                        const int q = it * NTHR + tid, row = q / CPRB, ch = q % CPRB;
                        const int rl = row % WNE, j = rl >> 4, a = (rl >> 2) & 3, b = rl & 3;
                        int n = n0 + (tr ? row : (row - rl) + 4 * NT * a + 4 * j + b);
                        n = n < p.N ? n : p.N - 1;
                        *reinterpret_cast<frag_t*>(bl + q * 16) = *reinterpret_cast<const frag_t*>(p.lora_b + ((int64_t)n * p.lora_r + c * LORA_RC) * (int)sizeof(T) + ch * 16);
                    }
                    __syncthreads();
                    bs = bl;
                }
#pragma unroll
                for (int ks = 0; ks < L_KS; ++ks) {
                    frag_t bf[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bf[j] = lds_read_frag(bs, (wn * WNE + 16 * j + c16) * RB + (4 * ks + g) * 16);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            if (tr) mma_step<T>(acc[i][j], tfr[i][ks], bf[j]);
                            else mma_step<T>(acc[i][j], bf[j], tfr[i][ks]);
                        }
                }
            }
        }
    }

    if constexpr (KG > 1) {
        // fixed-order sum of the two groups' partial tiles through LDS (the stage buffers are free once every wave is past
        // its last MFMA): group 1 deposits [wave][i][j][lane] float4s, group 0 adds them to its own and runs the epilogue
        __syncthreads();
        f32x4* ex = reinterpret_cast<f32x4*>(smem) + (wid * MT * NT) * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) ex[(i * NT + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 o = ex[(i * NT + j) * 64];
                acc[i][j][0] += o[0], acc[i][j][1] += o[1], acc[i][j][2] += o[2], acc[i][j][3] += o[3];
            }
    } else if (p.ln_stats) {
        __syncthreads();  // rowstat was written before the main loop by other waves; with max_kb >= 1 a barrier has passed, this covers total_kb == 0
    }

    if constexpr (LORA && !CONV) {
        if (ttile) {
            // ---- t-tile epilogue: what a producer publishes (t, or t / rstd = (x A'^T - mean sA) + cA / rstd with LayerNorm folded in), rounded to T,
            // written through to L2; then the flags of the tile's 32-row blocks, for every group ----
            constexpr int RUN = 4 * NT;
            const int nv = wn * WNE + RUN * g;  // this lane's first virtual column: RUN consecutive ranks of ONE group (R >= 32 >= RUN)
            const int rsh = p.lora_r == 32 ? 5 : p.lora_r == 64 ? 6 : 7;
            const int gi = nv >> rsh, r0 = nv & (p.lora_r - 1);
            const float* lsl = reinterpret_cast<const float*>(lora_b0);  // [groups x R] sA, then at + 256 floats cA (staged by the prologue)
            if (gi < p.lora_groups) {
                char* tg = p.lora_t + gi * p.lora_gs;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int mrow = wm * WME + 16 * i + c16, m = m0 + mrow;
                    if (m >= p.M) continue;
                    float v[RUN];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
                    if (p.ln_stats) {
                        const float mean = rowstat[2 * mrow], inv = 1.0f / rowstat[2 * mrow + 1];
#pragma unroll
                        for (int c = 0; c < RUN / 4; ++c) {
                            const f32x4 sa = *reinterpret_cast<const f32x4*>(lsl + gi * p.lora_r + r0 + 4 * c);
                            const f32x4 ca = *reinterpret_cast<const f32x4*>(lsl + 256 + gi * p.lora_r + r0 + 4 * c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 * c + e] = (v[4 * c + e] - mean * sa[e]) + ca[e] * inv;
                        }
                    }
                    char* dst = tg + ((int64_t)m * p.lora_r + r0) * (int)sizeof(T);
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int c = 0; c < RUN / 2; ++c) st_agent8(dst + 8 * c, __builtin_bit_cast(uint64_t, f32x2{v[2 * c], v[2 * c + 1]}));
                    } else {
#pragma unroll
                        for (int c = 0; c < RUN / 4; ++c) {
                            const bf16x4 b4 = {(bf16_t)v[4 * c], (bf16_t)v[4 * c + 1], (bf16_t)v[4 * c + 2], (bf16_t)v[4 * c + 3]};
                            st_agent8(dst + 8 * c, __builtin_bit_cast(uint64_t, b4));
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
            __syncthreads();
            constexpr int FB = BM / LORA_PM;
            if (tid < p.lora_groups * FB) {
                const int nfl = (p.M + LORA_PM - 1) / LORA_PM, fg = tid / FB, fb = m0 / LORA_PM + tid % FB;
                if (fb < nfl) __hip_atomic_store(p.lora_flags + fg * nfl + fb, tt_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }

    tile_epilogue<T, MT, NT, BM, CONV>(p, acc, rowstat, m0, n0, wm, wn, lane, tr, split);
}

template <typename T, int BM, int BN, int WM, int WN, bool CONV, int NSTAGE, int KG = 1, bool LORA = false>
int launch_cfg(const GemmP& p, hipStream_t stream) {
    constexpr int LDS0 = KG * NSTAGE * (BM + BN) * 128 + BM * 8 + (LORA ? BN * LORA_RC * (int)sizeof(T) : 0);
    constexpr int LDS = LDS0;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(KG == 1 || (BM / WM / 16) * (BN / WN / 16) * WM * WN * 1024 <= KG * NSTAGE * (BM + BN) * 128, "partial-tile exchange must fit the stage buffers");
    auto kfn = gemm_kernel<T, BM, BN, WM, WN, CONV, NSTAGE, KG, LORA>;
    static bool attr_set[64] = {};  // per device: the attribute belongs to the device's copy of the code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set[dev] = true;
    }
    GemmP q = p;
    plan_grid(q, BM, BN, CONV, KG);
    q.lora_dbg = g_lora_dbg;
    q.lora_tt = 0;
    if constexpr (LORA && !CONV && KG == 1) {
        const int mode = (g_lora_dbg & 64) ? 0 : (g_lora_dbg & 128) ? 2 : MI355X_LORA_TT;
        if (q.lora_groups * q.lora_r <= BN && q.ksplit <= 1 && q.nseg == 1 && (mode == 2 || (mode == 1 && q.lora_groups > 1))) q.lora_tt = 1;
    }
    // LoRA producers (or t-tiles), ahead of every output tile in dispatch order
    q.lp_blocks = !LORA ? 0 : q.lora_tt ? (q.tiles_m + 7) / 8 * 8 : ((q.M + LORA_PM - 1) / LORA_PM * q.lora_groups + 7) / 8 * 8;
    const int grid = q.pf_blocks + q.lp_blocks + q.grid0 * (q.ksplit > 1 ? q.ksplit : 1);
    // the (mean, rstd) rows are only allocated for launches that use them (64 x 64 tiles: 32 KB + 512 B would cost the fifth resident workgroup)
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WM * WN * 64 * KG), q.ln_stats ? LDS : LDS - BM * 8, stream, q);
    if (q.ksplit > 1) {
        const int rb = ((q.M + 31) / 32) * ((q.N + 63) / 64);
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(rb), dim3(256), 0, stream, q);
    }
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

// Tile configurations:  1: 128x128   2: 128x64   3: 64x128   4: 64x64   (4 waves, 2 x 2)
//                       6: 128x128 with two K groups (8 waves, intra-workgroup split-K)
// (5: 256x128 with 8 lockstep waves and 7 / 8: the same tile with its two 4-wave groups one barrier slot apart were measured level with
//  or behind two co-resident 128x128 workgroups on every shape of the step -- profiles/r02_i_probe_tiles.log, r02_g_autotune_merged.log --
//  and removed in round 3.  Round 4 re-tried the one-workgroup-per-CU idea with 128 x 64 WAVE tiles (fewer LDS reads per MFMA): 256x128 and
//  128x256 with 4 waves and a 3-deep ring, 256x256 with 8 waves -- behind the 128x128 tile on every shape of the step, level on a hot 4096^3
//  (profiles/r04_a_probe_tiles.log, which is also the hipBLASLt yardstick of this core); not kept.)
// The UNet's GEMMs are small for a 256-CU chip (2048x1280 outputs = 160 tiles of 128x128), so the choice is driven by
// how many workgroups a configuration yields: big tiles reuse operands better, small tiles fill the machine.  The engine
// passes measured choices per shape (refiners_amd/engine/tuning.py); this heuristic is the fallback.
inline int pick_tile(const GemmP& p, bool conv) {
    if ((g_tile >= 1 && g_tile <= 4) || g_tile == 6) return g_tile;
    if ((p.tile_hint >= 1 && p.tile_hint <= 4) || p.tile_hint == 6) return p.tile_hint;
    const int64_t b128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (conv) return 3;  // 64 x 128 wins for every conv shape of the UNet (r01_b probe: 339 / 540 / 570 TF at 32^2 / 64^2 / 128^2)
    if (p.geglu) return 1;
    if (b128 <= 256) return 4;
    if (b128 < 1000) return 2;
    return 1;
}
inline int pick_stages(const GemmP& p) {
    if (g_stages >= 2 && g_stages <= 4) return g_stages;
    if (p.stage_hint >= 2 && p.stage_hint <= 4) return p.stage_hint;
    return 2;  // two LDS stages by default: deeper pipelines cost a resident workgroup per CU
}

template <typename T, int BM, int BN, bool CONV>
int launch_stages(const GemmP& p, int stages, hipStream_t stream) {
    switch (stages) {
        case 2: return launch_cfg<T, BM, BN, 2, 2, CONV, 2>(p, stream);
        case 4: return launch_cfg<T, BM, BN, 2, 2, CONV, 4>(p, stream);
        default: return launch_cfg<T, BM, BN, 2, 2, CONV, 3>(p, stream);
    }
}

template <typename T, bool CONV>
int launch_tile(const GemmP& p, hipStream_t stream) {
    int tile = pick_tile(p, CONV);
    if (p.geglu && (tile == 2 || tile == 4)) tile = 3;  // the GEGLU epilogue needs 64 packed columns per wave
    if (p.colstats && tile == 6) tile = 1;               // column statistics come from the 4-wave tiles' epilogue (whoever asked for tile 6: hint, table or set_option)
    const int st = pick_stages(p);
    if (p.lora_b) {  // in-launch LoRA: the 4-wave tiles, two LDS stages; a stacked rank above 64 needs the 128-column tiles (the producers stage R weight rows)
        if (p.lora_r > 64 && (tile == 2 || tile == 4)) tile = tile == 2 ? 1 : 3;
        if constexpr (CONV) {
            return tile == 1 ? launch_cfg<T, 128, 128, 2, 2, true, 2, 1, true>(p, stream) : launch_cfg<T, 64, 128, 2, 2, true, 2, 1, true>(p, stream);
        } else {
            switch (tile) {
                case 2: return launch_cfg<T, 128, 64, 2, 2, false, 2, 1, true>(p, stream);
                case 3: return launch_cfg<T, 64, 128, 2, 2, false, 2, 1, true>(p, stream);
                case 4: return launch_cfg<T, 64, 64, 2, 2, false, 2, 1, true>(p, stream);
                default: return launch_cfg<T, 128, 128, 2, 2, false, 2, 1, true>(p, stream);
            }
        }
    }
#ifdef MI355X_PROBE_T10  // (probing build, round 6: 128 x 96 tiles computed by four waves stacked along M -- wave tile 32 x 96 -- for plain launches: does the layout that a
                         //  256-workgroup tile of N = 1280 (128 x 80) would need hold up against 64 x 64 tiles?  tools/probe_t10.py)
    if constexpr (!CONV) {
        if (g_tile == 10 || p.tile_hint == 10) return launch_cfg<T, 128, 96, 4, 1, false, 2>(p, stream);
        if (g_tile == 11 || p.tile_hint == 11) return launch_cfg<T, 128, 96, 4, 1, false, 3>(p, stream);
        // TWO-wave workgroups (128 threads): fewer LDS fragment reads per MFMA at the same tile -- 64 x 64 as 2 x (32 x 64): 0.75 instead of 1.0;
        // 128 x 64 as 2 x (64 x 64): 0.5 instead of 0.75 -- and more workgroups per CU for the same LDS
        if (g_tile == 12 || p.tile_hint == 12) return launch_cfg<T, 64, 64, 2, 1, false, 2>(p, stream);
        if (g_tile == 13 || p.tile_hint == 13) return launch_cfg<T, 64, 64, 2, 1, false, 3>(p, stream);
        if (g_tile == 14 || p.tile_hint == 14) return launch_cfg<T, 128, 64, 2, 1, false, 2>(p, stream);
        if (g_tile == 15 || p.tile_hint == 15) return launch_cfg<T, 128, 64, 2, 1, false, 3>(p, stream);
    }
#endif
    switch (tile) {
        case 1: return launch_stages<T, 128, 128, CONV>(p, st, stream);
        case 2: return launch_stages<T, 128, 64, CONV>(p, st, stream);
        case 3: return launch_stages<T, 64, 128, CONV>(p, st, stream);
        case 6: return launch_cfg<T, 128, 128, 2, 2, CONV, 2, 2>(p, stream);
        default: return launch_stages<T, 64, 64, CONV>(p, st, stream);
    }
}

int launch_conv_f32(const GemmP& p, hipStream_t stream);
int launch_conv_bf16(const GemmP& p, hipStream_t stream);

}  // namespace mi355x
