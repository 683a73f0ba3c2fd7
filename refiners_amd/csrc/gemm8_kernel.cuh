// gemm8_kernel.cuh: the second main loop of mi355x_gemm -- 256 x 256 output tile, 8 waves, EIGHT phases per two K tiles (gfx950).
//
//   out[M,N] = epi( sum_s X_s[M,K_s] . W_s[N,K_s]^T )        same contract, same LDS image, same epilogue as gemm_kernel.cuh
//
// Why a second loop: the 4-wave / two-phase loop of gemm_kernel.cuh has one barrier per K block with every wave of the workgroup in the
// same phase -- its waves sit on s_waitcnt / s_barrier half of their cycles (profiles/r04_g_pmc_sq.json) and no tile shape changed that
// (DESIGN.md section 4).  This loop is the CDNA programming guide's 256^2 template (cdna_hip_programming.md "The 256^2 8-phase template"):
//   * 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 outputs = acc[8][4] MMA tiles (128 registers), split into four 64 x 32 QUADRANTS;
//   * a phase = { LDS reads of the fragments one quadrant needs | LDS-DMA issue of ONE half tile of a later K tile } barrier { 16 MFMA of that
//     quadrant over the whole 128-byte K tile } barrier;  4 phases per K tile, the loop body covers two K tiles (LDS buffers 0 / 1);
//   * the two 4-wave halves (wm = 0 / 1) run ONE BARRIER APART: while one half issues its 16 MFMAs the other half reads LDS and issues
//     loads, so each SIMD (one wave of each half) always has a wave in the matrix pipe; s_setprio 1 around the MFMA cluster;
//   * vmcnt is counted, never 0 in steady state: waits only in phases 4 and 8 (`vmcnt(6)`: the three youngest half tiles stay in flight),
//     raw s_barrier (a __syncthreads() would drain the LDS-DMA queue);
//   * fragment registers: X quadrant rows (8 x 16 B) are re-read per half, both W halves (2 x 4 x 16 B) stay in registers for the K tile:
//     24 ds_read_b128 per wave and K tile for 64 MFMA (the 128 x 128 / 4-wave tile: 32 for 64).
// Half tiles (what one phase stages = 128 rows x 128 B = 2 loads per thread):
//   X half h = rows { 128 wm + 64 h + r }: the rows quadrant row h of BOTH wave rows;   W half h = LDS rows { 64 wn + 32 h + r } likewise.
// Per iteration (K tiles t -> buffer 0, t + 1 -> buffer 1), phase: reads | stage:
//   1: W0 X0 (buf 0) | X1 of t+1 -> buf 1      5: W0 X0 (buf 1) | X1 of t+2 -> buf 0
//   2: W1            | W0 of t+2 -> buf 0      6: W1            | W0 of t+3 -> buf 1
//   3: X1            | X0 of t+2 -> buf 0      7: X1            | X0 of t+3 -> buf 1
//   4: --            | W1 of t+2 -> buf 0 ; vmcnt(6): tile t+1 complete     8: -- | W1 of t+3 -> buf 1 ; vmcnt(6): tile t+2 complete
// Hazards (guide, same section): a half tile is READ one phase after the wait that retires it or later (the other half of the workgroup
// passes that wait one barrier later); a buffer is RE-STAGED two phases after its last ds_read, or one phase after when an lgkmcnt in front of
// the reading phase's first barrier retired those reads (phase 1 / 5: `lgkmcnt(8)` retires the four W0 reads, which are issued first).
// Loader: buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor): per-lane 32-bit offset that is constant over the K loop, the
// running K offset in the instruction's SCALAR offset, the descriptor per segment -- no vector ALU work per load -- and out-of-range lanes
// (conv padding, rows beyond M / N) are given offset 0x80000000, for which the hardware writes zeros to LDS (no zero page, no select).
// In-launch LoRA (its own kernel instance, LORA = true: the plain instance's register allocation is untouched): ONE column group of a plain one-segment GEMM
// without a transposed part.  t = x A^T comes from PRODUCER workgroups at the head of the grid -- gemm_lora_producer.cuh, the 4-wave kernel's producers: one per
// 32 rows, an 8-deep LDS ring in this launch's 128 KB of stage buffers, waves 4 .. 7 of the workgroup exit at once; t with the LayerNorm correction folded in,
// rounded to the storage type, write-through stores, one flag per 32 rows -- and every output tile adds T(t) (s B)^T after its K loop: fragments of t and of the
// up rows straight from memory (a wave's 16 rows x 64 B are 1 KB contiguous), 32 MFMAs per 32 ranks.  (Round 5 computed t in "t-tiles": the first tiles_m
// workgroups ran this loop on their row tile with the stacked down rows in the W slot.  t was then published when the tiles of the first dispatch round LEFT
// their K loops -- every one of them waited for the t-tile's epilogue and the hand-over, ~7 us of the 11 us the live LoRAs cost the CFG pair's FF1 -- where a
// producer's 20-K-block loop is done a quarter into the tiles' loop: profiles/r06_*.)  Register notes that shaped the code: (b) everything after the K loop
// reads the launch arguments through a pointer to the kernel-argument segment made opaque AFTER the loop: as fields of the by-value argument they are loaded at
// kernel entry, spilt over the loop and re-loaded one v_readlane per use (3 181 of them; 59 per 16-row block of the epilogue).
// Not in this loop (the host keeps such launches on gemm_kernel.cuh): other LoRA forms (several column groups, a transposed group, convolutions), operands of 2 GB and more.
#pragma once
#include <vector>

#include "gemm_epilogue.cuh"
#include "gemm_lora_producer.cuh"

namespace mi355x {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

MI_DEV rsrc_t make_rsrc(const char* base, int64_t bytes) {
    // wave-uniform by construction; readfirstlane makes that provable (guide T20: otherwise every load sits in a waterfall loop)
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, n, 0x00020000);
}
// 16 bytes per lane, global -> LDS: lane i lands at lds_wave_base + 16 i; source = descriptor base + voff (per lane) + soff (scalar)
MI_DEV void blds16(rsrc_t rs, char* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

// Workgroup -> tile (the XCD-aware rasterisation planned by plan_grid)
MI_DEV void tile_coords(const GemmP& p, int bx, int& tm, int& tn) {
    if (p.pn > 0) {  // rectangular regions (exact split of the tile grid, grid0 = 8 * hm * hn)
        const int xcd = bx & 7, idx = bx >> 3;
        const int rm = xcd / p.pn, rn = xcd - rm * p.pn;
        const int lm = idx / p.hn, ln = idx - lm * p.hn;
        tm = rm * p.hm + lm;
        tn = rn * p.hn + ln;
    } else {  // contiguous chunk of the row-major (pn == 0) or column-major (pn == -1) tile order per XCD
        const int id = xcd_remap(bx, p.grid0);
        if (p.pn == 0) {
            tm = id / p.tiles_n;
            tn = id - tm * p.tiles_n;
        } else {
            tn = id / p.tiles_m;
            tm = id - tn * p.tiles_m;
        }
    }
}

// Prefetch role (GemmP::pf_ptr): the first pf_blocks workgroups touch every 64 bytes of the next launch's weights
// Tile -> (first row, first column) of a two-height launch (GemmP::mix_*), XCD-aware: tile t runs on XCD t % 8 (persistent workgroup b takes tiles b, b + 256, ... and
// producer / prefetch workgroups come in multiples of 8).  Every XCD owns cb8 = mix_cb / 8 "early" column tiles and ca8 = (tiles_n - mix_cb) / 8 "late" ones:
//   192-row tiles (t < mix_nbig):  the XCD's early column tiles x the mix_rb / 192 row tiles of rows [0, mix_rb);
//   128-row tiles, first the rows [mix_rb, M) under the SAME early column tiles (their W panels are in this XCD's L2 from the 192-row tiles),
//                  then the XCD's late column tiles over ALL rows in 128-row tiles (one new W panel per late column tile).
MI_DEV void mix_coords(const GemmP& p, int t, int& m0, int& n0) {
    const int cb8 = p.mix_cb >> 3;
    if (t < p.mix_nbig) {
        const int x = t & 7, idx = t >> 3;
        const int tm = idx / cb8;
        m0 = tm * 192;
        n0 = (cb8 * x + (idx - tm * cb8)) * 256;
        return;
    }
    const int u = t - p.mix_nbig, x = u & 7, idx = u >> 3;
    const int nb8 = ((p.M - p.mix_rb) >> 7) * cb8;
    if (idx < nb8) {
        const int tm = idx / cb8;
        m0 = p.mix_rb + tm * 128;
        n0 = (cb8 * x + (idx - tm * cb8)) * 256;
    } else {
        const int j = idx - nb8, ca8 = (p.tiles_n - p.mix_cb) >> 3;
        const int tm = j / ca8;
        m0 = tm * 128;
        n0 = (p.mix_cb + ca8 * x + (j - tm * ca8)) * 256;
    }
}

template <int NTHR_ALL> MI_DEV void prefetch_role(const GemmP& p, int tid_all) {
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
}

#ifndef MI355X_G8_ABL
#define MI355X_G8_ABL 0  // probing builds only (results are wrong): bit 0 = no tile epilogue (the accumulators are kept alive, nothing is stored)
#endif
#ifndef MI355X_G8_PRIO
#define MI355X_G8_PRIO 1  // s_setprio 1 around every MFMA cluster (guide T5: +21..39 % on this schedule)
#endif

template <typename T, bool CONV, bool LORA, int MTK, int MT2 = 0>
__global__ __launch_bounds__(512) void gemm8_kernel(const GemmP p) {
    // MT2 != 0 (round 6): a launch of TWO tile heights -- tiles [0, p.mix_nbig) are 32 MTK rows high, the rest 32 MT2: FF1 of a CFG pair (2048 x 10240) is exactly 256 tiles of
    // 192 x 256 + 256 tiles of 128 x 256, one of each per CU, where 440 tiles of 192 rows are 1.72 dispatch rounds paid as two (mix_coords below).
    // The body of a segment is a generic lambda over the tile height; the LDS partition (stage buffers, epilogue vectors, spare area) is the larger tile's.
    // MT = 16-row blocks per wave: 8 = the 256 x 256 tile of the header; 6 = a 192 x 256 tile (wave tile 96 x 64, 12 MFMAs per phase) for launches whose 256-row tiles
    // leave CUs idle in their only dispatch round (M = 8192, N = 1280: 160 tiles of 256 rows on 256 CUs, 215 of 192 rows).  Everything row-related below is written in
    // WR = rows per wave row (128 / 96) and QR = rows of one X half tile per wave row (64 / 48); the W side does not change.
    static_assert(MTK == 8 || MTK == 6 || MTK == 4, "wave tile rows");  // (MT = 4: 128 x 256 tiles, wave tile 64 x 64, 8 MFMAs per phase -- tile id 10, round 6: the second round of a two-height FF1)
    static_assert(MT2 == 0 || (MT2 == 4 && MTK == 6 && !CONV), "two-height launches: 192-row + 128-row tiles of a plain GEMM");
    constexpr int BN = 256, NTHR = 512, NT = 4;
    constexpr int BUFB = (32 * MTK + BN) * 128;  // bytes of one LDS stage buffer (X tile + W tile) of the LARGER tile: the launch's LDS partition
    constexpr uint32_t OOB = 0x80000000u;                  // per-lane offset beyond every descriptor's num_records: the load writes zeros
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // per-tile vectors of the epilogue, TWO sets used alternately by consecutive segments of a persistent workgroup: the waves that finish a tile's epilogue first
    // go on to the next tile's prologue (which writes the other set) without waiting for the rest -- they meet again at the prologue's own barrier
    float* const epi_lds = reinterpret_cast<float*>(smem + 2 * BUFB);
    constexpr int EPI_SET = 2 * 256 + 2 * BN;  // floats: rowstat [BM][2] (mean, rstd) of the tile's rows (LayerNorm consumer) | colvec [2][BN]: bias (as float32) or folded-LayerNorm s | c
    int epi_par = 0;
    char* const spare = smem + 2 * BUFB + 2 * EPI_SET * 4;  // 2 KB nobody reads: where the two waves without X rows (MT = 6) let their zero-filling loads land

    const int tid0 = threadIdx.x, wid = wave_id();
    const int wm = wid >> 2, wn = wid & 3;
    if ((int)blockIdx.x < p.pf_blocks) {
        prefetch_role<NTHR>(p, tid0);
        return;
    }
    int bid = (int)blockIdx.x - p.pf_blocks;
    // In-launch LoRA (one column group, GemmP::lora_*; fluxion/adapters/lora.py:383-397): the first lp_blocks workgroups are PRODUCERS of t = x A^T (one per 32 rows:
    // LayerNorm correction folded in, rounded to the storage type like the reference's intermediate tensor, write-through stores, one flag per block).  Every output
    // tile adds T(t) (s B)^T after its K loop (32 MFMAs per wave and 32 ranks, t and the up rows read straight into fragment layout: no LDS) once its rows' flags
    // carry the launch's epoch.  Producers have the lowest ids: dispatched before any tile that will wait for them.
    if (LORA && p.lp_blocks > 0) {
        if (bid < p.lp_blocks) {  // producer role (gemm_lora_producer.cuh): a producer is 256 threads and one 32-row block of t
            // ranks 32 / 64: TWO producers per workgroup (waves 0-3 / 4-7, row blocks 2 bid / 2 bid + 1, half of the stage buffers each) -- a workgroup of this
            // launch owns a whole CU's LDS, so M / 32 one-producer workgroups would be a dispatch round of their own at M = 8192; rank 128 (20 KB per
            // ring stage): one producer with the whole ring, waves 4 .. 7 exit (an ended wave no longer counts at the workgroup's barriers)
            const int npb = (p.M + LORA_PM - 1) / LORA_PM, halves = p.lora_r <= 64 ? 2 : 1, half = wid >> 2;
            const int q = halves * bid + half;
            if (half >= halves || q >= npb || (p.lora_dbg & 1)) return;  // (padding up to a multiple of 8 keeps tile b on XCD b % 8; lora_dbg bit 0, probing / tests:
                                                                         //  the producers exit at once -- a LOST hand-over: every tile waits its 2 s and raises the launch's error word)
            constexpr int RING = 2 * BUFB, HALF = RING / 2, PB2 = (LORA_PM + 64) * 128, PB4 = (LORA_PM + 128) * 128;
            if (p.lora_r == 32) lora_producer<T, false, 1, (HALF / 8192 < 8 ? HALF / 8192 : 8)>(p, q, tid0 & 255, smem + half * HALF);
            else if (p.lora_r == 64) lora_producer<T, false, 2, (HALF / PB2 < 8 ? HALF / PB2 : 8)>(p, q, tid0 & 255, smem + half * HALF);
            else lora_producer<T, false, 4, (RING / PB4 < 8 ? RING / PB4 : 8)>(p, q);
            return;
        }
        bid -= p.lp_blocks;
    }
    // ---- this workgroup's span of the launch's work.  The unit of work is one K tile of one output tile; an output tile is sk_nk consecutive units.
    //   sk_mode 0: whole tiles: workgroup b takes tiles b, b + sk_g, ... in the XCD-aware rasterisation of plan_grid (sk_g = tiles: one each);
    //   sk_mode 1 ("stream-K"): sk_g persistent workgroups split the tiles x sk_nk units evenly (+-1) and contiguously, workgroup b taking span b.
    //   A span is walked BACKWARDS.  If it ends inside a tile, the workgroup first computes that tile's K tiles up to the span's end and DEPOSITS
    //   its float32 accumulators in sk_ws, raising flag b; then whole tiles; if it starts inside a tile, it finally computes that tile's LAST K
    //   range, COLLECTS what workgroups b - 1, b - 2, ... deposited for the earlier ranges (in that fixed order: the sum depends on the
    //   decomposition, not on timing -- bit-reproducible) and runs the tile's epilogue.  A deposit is the first thing a workgroup does, a collect
    //   the last, and an owner only ever waits for workgroups with LOWER ids, which the dispatcher started before it (no deadlock, whatever else
    //   shares the GPU).  Flags are zero between launches: the owner clears what it consumed.
    const int nk_tile = p.sk_nk;
    int64_t u0 = 0, u1 = 0;
    if (p.sk_mode == 1) {
        const int64_t U = (int64_t)p.grid0 * nk_tile;
        u0 = U * bid / p.sk_g;
        u1 = U * (bid + 1) / p.sk_g;
    }
    int tile_next = bid;  // sk_mode 0: whole tiles bid, bid + sk_g, bid + 2 sk_g, ... (sk_g = the number of workgroups; = the number of tiles for one tile each)

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    auto bar = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_barrier(); };
    auto lgkm0 = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    int ts_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr ((MI355X_G8_ABL & 4) != 0) {
            if (bid == 0 && tid0 == 0 && p.sk_ws && ts_n < 256) reinterpret_cast<uint64_t*>(p.sk_ws)[ts_n] = wall_clock64();
            ++ts_n;
        }
    };
    while (true) {
        stamp();  // (0) segment start
        // (the thread id goes through an opaque move once per segment: nothing derived from it -- epilogue addresses, hand-over offsets -- can be
        //  hoisted out of this loop and kept in registers through the K loop, which has none to spare)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        float* rowstat = epi_lds + epi_par * EPI_SET;
        float* colvec = rowstat + 2 * 256;
        epi_par ^= 1;
        const int lane = tid & 63, g = lane >> 4, c16 = lane & 15;
        // ---- loader geometry.  Half tile (h), load (s): 128 rows x 8 chunks = 1024 pieces of 16 B = 2 loads x 512 threads; piece s * 512 + tid is
        // (local row lr = s * 64 + 8 wid + (lane >> 3), physical chunk lane & 7).  X: LDS row = 128 s + 64 h + (lr & 63);  W: LDS row = 64 (lr >> 5)
        // + 32 h + (lr & 31).  A wave's 64 pieces are 8 consecutive LDS rows = 1 KB, lane-linear.  The bank swizzle of row R is (R >> 1) & 7, which
        // for every one of these rows equals 4 (wid & 1) + (lane >> 4): ONE source-chunk offset per thread.
        const int lr8 = lane >> 3;
        const uint32_t coff = (uint32_t)(((lane & 7) ^ (4 * (wid & 1) + (lane >> 4))) << 4);
        // row 16 q + c16 of a tile, logical chunk 4 kk + g: byte offset 128 (16 q) + fo[kk]
        int fo[2];
    #pragma unroll
        for (int kk = 0; kk < 2; ++kk) fo[kk] = c16 * 128 + (((4 * kk + g) ^ ((c16 >> 1) & 7)) << 4);
        int tile, kfirst = 0, nk = nk_tile;
        int64_t tbeg = 0;
        bool owner = true, more;
        if (p.sk_mode == 1) {
            if (u0 >= u1) break;
            tile = (int)((u1 - 1) / nk_tile);
            tbeg = (int64_t)tile * nk_tile;
            kfirst = u0 > tbeg ? (int)(u0 - tbeg) : 0;  // first K tile of this segment
            nk = (int)(u1 - tbeg) - kfirst;             // K tiles of this segment
            owner = kfirst + nk == nk_tile;             // this segment holds the tile's last K range: it ends with the tile's epilogue
            u1 -= nk;
            more = u0 < u1;
        } else {
            if (tile_next >= p.grid0) break;
            tile = tile_next;
            tile_next += p.sk_g;
            more = tile_next < p.grid0;
        }
        const bool small_tile = MT2 != 0 && p.mix_nbig > 0 && tile >= p.mix_nbig;
        auto segment = [&](auto mtc) __attribute__((always_inline)) {
        constexpr int MT = decltype(mtc)::value;
        constexpr int BM = 32 * MT, WR = 16 * MT, QR = 8 * MT, XW = QR / 8;  // XW = waves that stage an X half tile (8 rows each)
        constexpr int XB = BM * 128;                                          // bytes of the X tile inside a stage buffer (the buffers keep the larger tile's pitch BUFB)
        auto lgkm8 = [&]() __attribute__((always_inline)) {  // the X reads of the phase (MT: issued behind the four W reads) may stay outstanding, the W reads may not
            if constexpr (MT == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else if constexpr (MT == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        };

        int tm = 0, tn = 0, m0, n0;
        if (MT2 != 0 && p.mix_nbig > 0) {
            mix_coords(p, tile, m0, n0);
        } else {
            if (p.sk_mode == 1) {  // tiles in row-major (sk_order 0) or column-major order along the unit axis
                if (p.sk_order == 0) {
                    tm = tile / p.tiles_n;
                    tn = tile - tm * p.tiles_n;
                } else {
                    tn = tile / p.tiles_m;
                    tm = tile - tn * p.tiles_m;
                }
            } else {
                tile_coords(p, tile, tm, tn);
            }
            m0 = tm * BM, n0 = tn * BN;
        }

        // transposed tile (columns >= nt_begin, stored as out_t[n][m]): the tile of the TRANSPOSED problem -- the LDS slot that normally holds activation
        // rows is filled from the weight rows n0 + R, the slot that normally holds (permuted) weight rows from the activation rows m0 + perm(R).  The K
        // loop does not know; acc[i][j] then is the (weight rows 16 i.., activation rows 16 j..) block, and the epilogue gets its transpose.
        const bool tr = !CONV && n0 >= p.nt_begin;
        const int xs_lim = tr ? p.N : p.M, ws_lim = tr ? p.M : p.N, xs_0 = tr ? n0 : m0, ws_0 = tr ? m0 : n0;
        // Source rows of the two slots' LDS rows, as one lane constant each (the per-(h, s) part is a compile-time offset):
        //   X slot: row 128 s + 64 h + xlane of the tile;   W slot: LDS row R = 64 (2 s + (wid >> 2)) + 32 h + q, q = 8 (wid & 3) + (lane >> 3), holds source
        //   row (R - rl) + 16 a + 4 j + b with rl = R % 64 = 16 j + 4 a + b (gemm_kernel.cuh header: every lane then owns 16 consecutive output columns)
        //   = 128 s + 8 h + wlane.  Rows beyond the operand get row -1 (offset 2^32 - ld + coff: beyond every descriptor, the load writes zeros).
        const int wq = 8 * (wid & 3) + lr8;
        const int xlane = 8 * wid + lr8, wlane = 64 * (wid >> 2) + 16 * ((wq >> 2) & 3) + 4 * (wq >> 4) + (wq & 3);
        auto xrow = [&](int h, int s2) __attribute__((always_inline)) {
            const int r = xs_0 + WR * s2 + QR * h + xlane;
            return r < xs_lim && (MT == 8 || wid < XW) ? r : -1;  // (MT = 6: waves 6 and 7 stage nothing real -- zeros into a spare LDS area)
        };
        auto wrow = [&](int h, int s2) __attribute__((always_inline)) {
            const int r = ws_0 + 128 * s2 + 8 * h + wlane;
            return r < ws_lim ? r : -1;
        };
        int xb[2][2], xyx[2][2];  // conv: image index (-1: a row beyond M), (oy | ox << 16) of the output pixel
        if constexpr (CONV) {
            const int ohw = p.OH * p.OW;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int mr = xrow(h, s2), m = mr < 0 ? 0 : mr;
                    const int bi = m / ohw, rem = m - bi * ohw, oy = rem / p.OW;
                    xb[h][s2] = mr < 0 ? -1 : bi;
                    xyx[h][s2] = oy | ((rem - oy * p.OW) << 16);
                }
        }

        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- the segment's position in the K segments of the launch ----
        int seg0 = 0, kb0 = kfirst;
        while (seg0 < p.nseg - 1 && kb0 >= p.seg[seg0].nkb) {
            kb0 -= p.seg[seg0].nkb;
            ++seg0;
        }

        // ---- two independent cursors (the X halves of a K tile are staged in other phases than its W halves) ----
        uint32_t xvo[2][2], wvo[2][2];  // per-lane byte offsets into the current segment's x / w
        rsrc_t xrs, wrs;
        int x_seg = seg0, x_kb = kb0, x_nkb = 0, x_cpb = 1, x_cb = 0, x_tap = 0;
        int w_seg = seg0, w_kb = kb0, w_nkb = 0;
        uint32_t x_so = 0, x_step = 128, w_so = 0, w_step = 128;
        int c_ks = 1, c_pad = 0, c_st = 1, c_up = 0, c_H = 1, c_W = 1;  // conv: the current segment's geometry, held in scalars (set_tap runs inside the K loop)
        uint32_t c_ld = 0;
        auto set_tap = [&]() __attribute__((always_inline)) {
            int dy = x_tap / c_ks, dx = x_tap - dy * c_ks;
            dy -= c_pad;
            dx -= c_pad;
            const int HH = c_H << c_up, WW = c_W << c_up;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int iy = (xyx[h][s] & 0xffff) * c_st + dy, ix = (xyx[h][s] >> 16) * c_st + dx;
                    const bool ok = xb[h][s] >= 0 && iy >= 0 && iy < HH && ix >= 0 && ix < WW;
                    const int pix = (xb[h][s] * c_H + (iy >> c_up)) * c_W + (ix >> c_up);
                    xvo[h][s] = ok ? (uint32_t)pix * c_ld + coff : OOB;
                }
        };
        auto enter_x = [&](int s, int kb) __attribute__((always_inline)) {
            const SegP& sp = p.seg[s];
            x_nkb = sp.nkb;
            xrs = make_rsrc(sp.x, sp.xbytes);
            if constexpr (CONV) {
                x_cpb = sp.cpb;
                x_tap = kb / sp.cpb;
                x_cb = kb - x_tap * sp.cpb;
                x_so = (uint32_t)x_cb * 128u;
                c_ks = sp.ksize, c_pad = sp.pad, c_st = sp.stride, c_up = sp.ups_shift, c_H = sp.H, c_W = sp.W, c_ld = (uint32_t)sp.ldxb;
                set_tap();
            } else {
                const bool kbl = tr ? sp.wkb : sp.xkb;
                if (tr) xrs = make_rsrc(sp.w, sp.wbytes);
                x_step = kbl ? (uint32_t)xs_lim * 128u : 128u;
                x_so = (uint32_t)kb * x_step;
                const uint32_t ld = kbl ? 128u : (uint32_t)(tr ? sp.ldwb : sp.ldxb);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) xvo[h][s2] = (uint32_t)xrow(h, s2) * ld + coff;
            }
        };
        auto enter_w = [&](int s, int kb) __attribute__((always_inline)) {
            const SegP& sp = p.seg[s];
            w_nkb = sp.nkb;
            bool kbl = tr ? sp.xkb : sp.wkb;
            wrs = tr ? make_rsrc(sp.x, sp.xbytes) : make_rsrc(sp.w, sp.wbytes);
            w_step = kbl ? (uint32_t)ws_lim * 128u : 128u;
            uint32_t ld = kbl ? 128u : (uint32_t)(tr ? sp.ldxb : sp.ldwb);
            w_so = (uint32_t)kb * w_step;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) wvo[h][s2] = (uint32_t)wrow(h, s2) * ld + coff;
        };
        auto adv_x = [&]() __attribute__((always_inline)) {  // one K tile forward (called behind the stage of X half 1)
            ++x_kb;
            if constexpr (CONV) {
                x_so += 128u;
                if (++x_cb == x_cpb) {
                    x_cb = 0;
                    x_so = 0;
                    ++x_tap;
                    if (x_kb < x_nkb) set_tap();
                }
            } else {
                x_so += x_step;
            }
            if (x_kb == x_nkb) {
                x_kb = 0;
                if (++x_seg < p.nseg) enter_x(x_seg, 0);
            }
        };
        auto adv_w = [&]() __attribute__((always_inline)) {
            ++w_kb;
            w_so += w_step;
            if (w_kb == w_nkb) {
                w_kb = 0;
                if (++w_seg < p.nseg) enter_w(w_seg, 0);
            }
        };
        auto stage_x = [&](int h, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) blds16(xrs, MT == 8 || wid < XW ? smem + buf * BUFB + (WR * s + QR * h + 8 * wid) * 128 : spare + (wid - XW) * 1024, xvo[h][s], x_so);
        };
        auto stage_w = [&](int h, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) blds16(wrs, smem + buf * BUFB + XB + (64 * (2 * s + (wid >> 2)) + 32 * h + 8 * (wid & 3)) * 128, wvo[h][s], w_so);
        };

        // ---- fragments ----
        frag_t xf[MT / 2][2];  // X quadrant rows: [16-row block][K half]
        frag_t wf[2][2][2];  // W halves: [h][16-row block][K half]
        auto read_x = [&](int h, int buf) __attribute__((always_inline)) {
            const char* xs = smem + buf * BUFB + (WR * wm + QR * h) * 128;
#pragma unroll
            for (int i = 0; i < MT / 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) xf[i][kk] = lds_read_frag(xs, i * 2048 + fo[kk]);
        };
        auto read_w = [&](int h, int buf) __attribute__((always_inline)) {
            const char* ws = smem + buf * BUFB + XB + (64 * wn + 32 * h) * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) wf[h][j][kk] = lds_read_frag(ws, j * 2048 + fo[kk]);
        };
        auto mma_q = [&](auto hxc, auto hwc) __attribute__((always_inline)) {  // quadrant (hx, hw): 16 MMA steps
            constexpr int hx = decltype(hxc)::value, hw = decltype(hwc)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MT / 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma_step<T>(acc[(MT / 2) * hx + i][2 * hw + j], wf[hw][j][kk], xf[i][kk]);
        };
        // the compute half of a phase: barrier | the fragments have arrived | 16 MFMA at raised priority | barrier
        auto compute = [&](auto hxc, auto hwc, bool reads) __attribute__((always_inline)) {
            fence();
            bar();
            if (reads) lgkm0();
            fence();
            if constexpr (MI355X_G8_PRIO != 0) __builtin_amdgcn_s_setprio(1);
            mma_q(hxc, hwc);
            if constexpr (MI355X_G8_PRIO != 0) __builtin_amdgcn_s_setprio(0);
            fence();
            bar();
            fence();
        };

        // ---- prologue: K tile 0 completely, three half tiles of K tile 1 ----
        enter_x(x_seg, x_kb);
        enter_w(w_seg, w_kb);
        stage_w(0, 0);
        stage_x(0, 0);
        stage_w(1, 0);
        adv_w();
        stage_x(1, 0);
        adv_x();
        if (nk > 1) {
            stage_w(0, 1);
            stage_x(0, 1);
            stage_w(1, 1);
            adv_w();
        }
        if (p.ln_stats && owner) ln_rowstat<256, NTHR>(p, m0, tid, rowstat);  // (256 rows whatever the tile: two threads per row; a 192-row tile's last 64 are the next tile's, unused)
        if (owner && tid < BN && (p.ln_stats || p.bias)) {  // per-column vectors of the tile -> LDS (read by the epilogue: see tile_epilogue's colvec)
            const int n = min(n0 + tid, p.N - 1);
            if (p.ln_stats) {
                colvec[tid] = p.ln_s[n];
                colvec[BN + tid] = p.ln_c[n];
            } else {
                colvec[tid] = to_f32(reinterpret_cast<const T*>(p.bias)[n]);
            }
        }
        if (nk > 1) wait_vm<6>();
        else wait_vm0();
        stamp();  // (1) first K tile landed
        fence();
        bar();
        if (wm == 1) bar();  // the second half of the workgroup runs one barrier behind the first from here on
        fence();

        // ---- main loop: two K tiles per trip.  GUARD = false: tiles t .. t + 3 exist (no conditions anywhere); true: the last one or two trips ----
        auto trip = [&](auto guardc, int t) __attribute__((always_inline)) {
            constexpr bool G = decltype(guardc)::value;
            const bool e1 = !G || t + 1 < nk, e2 = !G || t + 2 < nk, e3 = !G || t + 3 < nk;
            // phase 1
            read_w(0, 0);
            fence();  // (the four W reads first: lgkmcnt(8) below counts on it)
            read_x(0, 0);
            if (e1) stage_x(1, 1);
            lgkm8();
            if (e1) adv_x();  // (behind the counted wait: a segment change issues scalar loads, which share lgkmcnt with the LDS reads)
            compute(I0{}, I0{}, true);
            // phase 2
            read_w(1, 0);
            if (e2) stage_w(0, 0);
            compute(I0{}, I1{}, true);
            // phase 3
            read_x(1, 0);
            if (e2) stage_x(0, 0);
            compute(I1{}, I1{}, true);
            // phase 4
            if (e2) {
                stage_w(1, 0);
                adv_w();
                wait_vm<6>();
            } else {
                wait_vm0();
            }
            compute(I1{}, I0{}, false);
            if (G && !e1) return;
            // phase 5
            read_w(0, 1);
            fence();
            read_x(0, 1);
            if (e2) stage_x(1, 0);
            lgkm8();
            if (e2) adv_x();
            compute(I0{}, I0{}, true);
            // phase 6
            read_w(1, 1);
            if (e3) stage_w(0, 1);
            compute(I0{}, I1{}, true);
            // phase 7
            read_x(1, 1);
            if (e3) stage_x(0, 1);
            compute(I1{}, I1{}, true);
            // phase 8
            if (e3) {
                stage_w(1, 1);
                adv_w();
                wait_vm<6>();
            } else {
                wait_vm0();
            }
            compute(I1{}, I0{}, false);
        };
        int t = 0;
        for (; t + 3 < nk; t += 2) trip(std::false_type{}, t);
        for (; t < nk; t += 2) trip(std::true_type{}, t);
        if (wm == 0) bar();  // (balances the extra barrier of the second half: every wave is past its last LDS read, every stage has landed)
        fence();
        // Every LDS-DMA has landed (the last trip waited vmcnt(0) in inline asm, which the compiler's wait-count pass does not parse).  Say so in a form
        // it does: otherwise it still believes LDS-DMA writes are pending and puts `s_waitcnt vmcnt(0)` in front of every LDS read of the epilogue
        // (row statistics, staged column vectors) -- and vmcnt counts the previous row's STORES: 8 store round trips per tile, 8 us
        // (profiles/r05_h_probe_g8_stamps.log).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) expcnt(7) lgkmcnt(15)
        stamp();  // (2) K loop done

        // ---- stream-K hand-over.  Slot layout: [workgroup][wave][i][j][lane] float4 -- a wave's 64 lanes write / read 1 KB contiguous ----
        if (!owner) {
            // deposit: write-through (sc1) stores, acknowledged (vmcnt(0)) by every wave, barrier, then the flag (guide: "sc1 slab stores ->
            // every wave s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope store"; the same hand-off form as the in-launch LoRA's t)
            const rsrc_t srs = make_rsrc(reinterpret_cast<const char*>(p.sk_ws) + (int64_t)bid * (BM * BN * 4), BM * BN * 4);
            const uint32_t so = (uint32_t)((wid * MT * NT) * 64 + lane) * 16u;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), srs, so + (uint32_t)((i * NT + j) * 1024), 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.sk_flags + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (kfirst > 0) {
            // collect: workgroups bid - 1, bid - 2, ... hold the K tiles [.., kfirst) of this tile
            const int64_t U = (int64_t)p.grid0 * nk_tile;
            int have = kfirst, src = bid;
            while (have > 0) {
                --src;
                const int64_t s0 = U * src / p.sk_g, s1 = U * (src + 1) / p.sk_g;
                have -= (int)(s1 - (s0 > tbeg ? s0 : tbeg));
                {   // every wave polls for itself (no cross-wave hand-off); relaxed agent-scope loads bypass the CU's L1
                    const uint64_t t0 = wall_clock64();
                    while (__hip_atomic_load(p.sk_flags + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) {
                        __builtin_amdgcn_s_sleep(2);
                        if (wall_clock64() - t0 > 200000000ull) {  // 2 s of the 100 MHz clock: a lost workgroup must not hang the GPU; the host finds the error word set
                            if (lane == 0) __hip_atomic_store(p.sk_flags + p.sk_cap, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                const rsrc_t srs = make_rsrc(reinterpret_cast<const char*>(p.sk_ws) + (int64_t)src * (BM * BN * 4), BM * BN * 4);
                const uint32_t so = (uint32_t)((wid * MT * NT) * 64 + lane) * 16u;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x4 part[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) part[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, so + (uint32_t)((i * NT + j) * 1024), 0, 16));
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] += part[j];
                }
            }
            __syncthreads();  // every wave has seen the flags: clear them for the next launch (the slots are free once the loads above returned)
            if (tid < bid - src) __hip_atomic_store(p.sk_flags + src + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (everything below derives its per-lane addresses from a copy of the lane id made opaque HERE: computed any earlier -- the compiler hoists such
        //  arithmetic above the K loop -- it would sit in registers the K loop does not have)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int ge = lane_e >> 4, ce = lane_e & 15;
        // (and the LoRA roles read their parameters through a pointer to the kernel arguments made opaque here, for the same reason: as fields of `p` they
        //  are scalar loads the compiler issues at kernel entry and carries -- spilt -- through the K loop)
        const __attribute__((address_space(4))) GemmP* pe = (const __attribute__((address_space(4))) GemmP*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pe));
#ifdef MI355X_G8_EPI_ARGS_AT_ENTRY  // (A/B build: the epilogue reads the by-value kernel argument like the K loop does: ~60 scalar-spill reloads per 16-row block)
        const GemmP& pq = p;
#else
        const GemmP& pq = *(const GemmP*)pe;  // (the epilogue's view of the arguments: loaded HERE, after the K loop, into scalar registers the loop no longer needs)
#endif
        if (LORA) {
            // ---- LoRA tail of an output tile: acc += T(t) (s B)^T, 32 ranks per step ----
            const int nfl = (pe->M + 31) / 32, lora_tag = *pe->lora_epoch;
            {  // this wave's four row blocks: lanes 0..3 poll one flag each (relaxed agent-scope loads bypass the CU's L1), bounded by the wall clock
                const int fb = (m0 + WR * wm) / 32 + min(lane_e & 3, WR / 32 - 1);
                const int* fp = pe->lora_flags + (fb < nfl ? fb : nfl - 1);
                const uint64_t t0 = wall_clock64();
                while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != lora_tag) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > 200000000ull) {  // 2 s: raise the launch's error word (native.LoraSync.check) and go on
                        __hip_atomic_store(pe->lora_flags + pe->lora_groups * nfl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            asm volatile("" ::: "memory");  // (no load of t may be moved above the poll: relaxed atomics order nothing for the compiler)
            constexpr int KS32 = 32 / DT<T>::KSTEP;  // MMA steps per 32 ranks (bf16: 1, f32: 2)
            const int rb = pe->lora_r * (int)sizeof(T);  // bytes per row of t / of the up rows
            // the W slot's row order (gemm_kernel.cuh header): MMA row ce of block j is output column 64 wn + 16 (ce >> 2) + 4 j + (ce & 3)
            const int ncol = n0 + 64 * wn + 16 * (ce >> 2) + (ce & 3);
            const int nc32 = pe->lora_r / 32;
            for (int c = 0; c < nc32; ++c) {
                frag_t tfr[MT][KS32], bfr[NT][KS32];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int m = min(m0 + WR * wm + 16 * i + ce, pe->M - 1);
#pragma unroll
                    for (int ks = 0; ks < KS32; ++ks)
                        tfr[i][ks] = *reinterpret_cast<const frag_t*>(pe->lora_t + (int64_t)m * rb + c * 32 * (int)sizeof(T) + (4 * ks + ge) * 16);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = min(ncol + 4 * j, pe->N - 1);
#pragma unroll
                    for (int ks = 0; ks < KS32; ++ks) bfr[j][ks] = *reinterpret_cast<const frag_t*>(pe->lora_b + (int64_t)n * rb + c * 32 * (int)sizeof(T) + (4 * ks + ge) * 16);
                }
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) mma_step<T>(acc[i][j], bfr[j][ks], tfr[i][ks]);
            }
        }
        if constexpr ((MI355X_G8_ABL & 1) != 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else if constexpr (!CONV) {
            if (MT == 8 && tr) {  // the transposed tile's blocks: 4 x 8 over (activation rows of wave column wn, weight rows of wave row wm)  (256-row tiles only: gemm8_ok)
                f32x4 at[NT][MT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) at[j][i] = acc[i][j];
                tile_epilogue<T, NT, MT, BM, false, true>(pq, at, rowstat, m0, n0, wn, wm, lane_e, true, 0, colvec);
            } else {
                tile_epilogue<T, MT, NT, BM, false, false, true>(pq, acc, rowstat, m0, n0, wm, wn, lane_e, false, 0, colvec);
            }
        } else {
            tile_epilogue<T, MT, NT, BM, true, false, true>(pq, acc, rowstat, m0, n0, wm, wn, lane_e, false, 0, colvec);
        }
        stamp();  // (3) epilogue issued
        // (no barrier between segments: the next prologue writes the OTHER set of epilogue vectors, and the stage buffers it fills were last read before
        //  the barrier that closed the K loop; the set this epilogue reads is rewritten two segments on, behind the next prologue's barrier)
#ifdef MI355X_G8_END_BARRIER  // (A/B build: the round-5 first version, every wave waits for the tile's last epilogue row before the next tile's first load)
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#endif
        };  // segment
        if constexpr (MT2 != 0) {
            if (small_tile) segment(std::integral_constant<int, MT2>{});
            else segment(std::integral_constant<int, MTK>{});
        } else {
            segment(std::integral_constant<int, MTK>{});
        }
    }
}

extern int g_sk_g;        // probing: number of stream-K / persistent workgroups (0 = one per CU)
extern int g_lora_dbg;    // probing bits of the in-launch LoRA (gemm.hip; bit 0 = the hand-over's producers exit at once)
extern int g_g8_persist;  // 1 = launches with more tiles than CUs run as one persistent workgroup per CU

// Geometry of a two-height launch (tile id 11) for n_cu CUs, or false: rows [0, rb) in 192-row tiles over the first cb column tiles = a whole number of rounds, everything else in
// 128-row tiles = a whole number of rounds, each XCD with whole column tiles of every region (mix_coords).
inline bool plan_mix(int M, int N, int n_cu, int& rb, int& cb, int& nbig, int& nsmall) {
    if (M <= 0 || N <= 0 || M % 128 || N % 256 || n_cu <= 0 || n_cu % 8) return false;
    const int tn = N / 256;
    if (tn % 8) return false;
    for (rb = M / 384 * 384; rb >= 384; rb -= 384) {
        if ((M - rb) % 128) continue;
        const int rows_big = rb / 192;
        for (cb = tn / 8 * 8 - 8; cb >= 8; cb -= 8) {
            nbig = rows_big * cb;
            nsmall = (rb / 128) * (tn - cb) + ((M - rb) / 128) * tn;
            if (nbig % n_cu == 0 && nsmall % n_cu == 0) return true;
        }
    }
    return false;
}

template <typename T, bool CONV, bool LORA, int MT, int MT2 = 0>
int launch_gemm8_impl(const GemmP& p, hipStream_t stream, bool streamk) {
    constexpr int LDS = 2 * (32 * MT + 256) * 128 + 2 * (256 * 8 + 2 * 256 * 4) + (MT == 4 || MT2 == 4 ? 4096 : 2048);  // two stage buffers + two sets of epilogue vectors + the spare landing area (1 KB per wave without X rows)
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kfn = gemm8_kernel<T, CONV, LORA, MT, MT2>;
    static bool attr_set[64] = {};
    static int n_cu[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev);
        attr_set[dev] = true;
    }
    GemmP q = p;
    plan_grid(q, 32 * MT, 256, CONV, 2);
    q.lora_dbg = g_lora_dbg & 1;  // (mi355x_set_option "lora_dbg"; the other probing bits belong to the 4-wave kernel's producers)
    q.lora_tt = 0;
    // producers: one per 32 rows, two to a workgroup for ranks 32 / 64 (see the kernel), padded to a multiple of 8 workgroups (tile b stays on XCD b % 8)
    q.lp_blocks = q.lora_b ? (((q.M + LORA_PM - 1) / LORA_PM + (q.lora_r <= 64 ? 1 : 0)) / (q.lora_r <= 64 ? 2 : 1) + 7) / 8 * 8 : 0;
    q.ksplit = 1;
    q.sk_nk = 0;
    for (int s = 0; s < q.nseg; ++s) q.sk_nk += q.seg[s].nkb;
    q.sk_mode = 0;
    q.sk_g = q.grid0;  // one tile per workgroup
    int ncu = dev >= 0 && dev < 64 && n_cu[dev] > 0 ? n_cu[dev] : 256;
    if (g_sk_g > 0) ncu = g_sk_g;
    if (streamk && !q.lora_b && q.sk_ws && q.sk_flags && q.grid0 % ncu != 0) {
        // "stream-K": one persistent workgroup per CU (256 on MI355X; the decomposition -- hence the summation order -- depends on this number only),
        // fewer when there is less than two K tiles of work for each
        int G = ncu;
        const int64_t U = (int64_t)q.grid0 * q.sk_nk;
        if (U < 2 * (int64_t)G) G = (int)(U / 2 > 0 ? U / 2 : 1);
        if (G > q.sk_cap) G = q.sk_cap;
        if (G > 0 && q.grid0 % G != 0) {
            q.sk_mode = 1;
            q.sk_g = G;
            // tile order along the unit axis: workgroup b runs on XCD b % 8; count, for both orders, the operand panels each XCD's L2 has to pull
            double kx = 0, kw = 0;
            for (int s = 0; s < q.nseg; ++s) {
                kw += q.seg[s].nkb;
                kx += CONV ? (double)q.seg[s].nkb / (q.seg[s].ksize * q.seg[s].ksize) : (double)q.seg[s].nkb;
            }
            double best = 0;
            for (int order = 0; order < 2; ++order) {
                std::vector<char> seen_m(8 * q.tiles_m, 0), seen_n(8 * q.tiles_n, 0);
                for (int b = 0; b < G; ++b) {
                    const int64_t s0 = U * b / G, s1 = U * (b + 1) / G;
                    for (int64_t t = s0 / q.sk_nk; t <= (s1 - 1) / q.sk_nk; ++t) {
                        const int tm = order == 0 ? (int)(t / q.tiles_n) : (int)(t % q.tiles_m), tn = order == 0 ? (int)(t % q.tiles_n) : (int)(t / q.tiles_m);
                        seen_m[(b & 7) * q.tiles_m + tm] = 1;
                        seen_n[(b & 7) * q.tiles_n + tn] = 1;
                    }
                }
                double cost = 0;
                for (char c : seen_m) cost += c ? kx : 0;
                for (char c : seen_n) cost += c ? kw : 0;
                if (order == 0 || cost < best) {
                    best = cost;
                    q.sk_order = order;
                }
            }
        }
    }
    q.mix_nbig = q.mix_rb = q.mix_cb = 0;
    if constexpr (MT2 != 0) {  // two tile heights: every CU one 192-row tile, then one 128-row tile (or whole multiples); the caller checked plan_mix
        int rb, cb, nbig, nsmall;
        if (!plan_mix(q.M, q.N, ncu, rb, cb, nbig, nsmall)) return MI355X_ESHAPE;
        q.mix_nbig = nbig, q.mix_rb = rb, q.mix_cb = cb;
        q.tiles_n = q.N / 256;
        q.grid0 = nbig + nsmall;
        q.sk_mode = 0;
        q.sk_g = ncu;
    }
    // several whole tiles per workgroup: b, b + ncu, ... (the same XCD each time).  LoRA launches too (round 6): their producers leave the CU after a quarter of a tile's
    // time, and the persistent workgroups that start behind them are the highest ids -- the ones with the fewest tiles
    if (q.sk_mode == 0 && g_g8_persist && q.grid0 > ncu && ncu % 8 == 0) q.sk_g = ncu;
    const int grid = q.pf_blocks + q.lp_blocks + q.sk_g;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS, stream, q);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

template <typename T, bool CONV>
int launch_gemm8(const GemmP& p, hipStream_t stream, bool streamk, int mt) {  // mt: 8 = 256-row tiles (tile ids 7 / 8), 6 = 192-row tiles (tile id 9), 4 = 128-row tiles (tile id 10: bf16 GEMMs only), 11 = 192- and 128-row tiles in one launch (tile id 11)
    if constexpr (!CONV && sizeof(T) == 2) {
        if (mt == 11) return p.lora_b ? launch_gemm8_impl<T, false, true, 6, 4>(p, stream, false) : launch_gemm8_impl<T, false, false, 6, 4>(p, stream, false);
        if (mt == 4) return p.lora_b ? launch_gemm8_impl<T, false, true, 4>(p, stream, false) : launch_gemm8_impl<T, false, false, 4>(p, stream, false);
    }
    if constexpr (!CONV) {
        if (p.lora_b) {  // (its own instance: the LoRA roles cost the plain one registers it does not have)
            return mt == 6 ? launch_gemm8_impl<T, false, true, 6>(p, stream, false) : launch_gemm8_impl<T, false, true, 8>(p, stream, streamk);
        }
    }
    return mt == 6 ? launch_gemm8_impl<T, CONV, false, 6>(p, stream, false) : launch_gemm8_impl<T, CONV, false, 8>(p, stream, streamk);
}

// Can this launch run on the 8-phase loop?  (No in-launch LoRA, no split-K workspace protocol, transposed column groups from a multiple of 256; every operand below
// 2 GB: 32-bit buffer offsets with 0x80000000 as the out-of-range marker.)
inline bool gemm8_ok(const GemmP& p, bool conv = false, int mt = 8) {
    if (mt != 8 && p.out_t) return false;  // (the 192- and 128-row tiles have no transposed form)
    if (mt == 4 && conv) return false;  // (tile id 10 is instantiated for bf16 GEMMs: gemm.hip keeps float32 launches off it)
    if (p.ksplit > 1 || !p.vec_ok || p.N % 16) return false;  // (the epilogue instances of this loop are the vectorised ones)
    if (p.lora_b && (conv || p.lora_groups != 1 || p.nseg != 1 || p.out_t || (p.lora_r != 32 && p.lora_r != 64 && p.lora_r != 128) || !p.lora_t || !p.lora_flags || !p.lora_epoch)) return false;  // in-launch LoRA here: one column group of a plain GEMM
    if (p.out_t && p.nt_begin % 256) return false;  // a tile is either stored row-major or transposed
    for (int s = 0; s < p.nseg; ++s)
        if (p.seg[s].xbytes <= 0 || p.seg[s].wbytes <= 0 || p.seg[s].xbytes >= (1ll << 31) || p.seg[s].wbytes >= (1ll << 31)) return false;
    return true;
}

int launch_gemm8_f32(const GemmP& p, hipStream_t stream, bool streamk, int mt);
int launch_gemm8_bf16(const GemmP& p, hipStream_t stream, bool streamk, int mt);
int launch_conv8_f32(const GemmP& p, hipStream_t stream, bool streamk, int mt);
int launch_conv8_bf16(const GemmP& p, hipStream_t stream, bool streamk, int mt);

}  // namespace mi355x
