// gemm_lora_producer.cuh: the producer role of mi355x_gemm's in-launch LoRA -- t = x A_cat^T for one (column group, 32-row block) -- shared by the two GEMM
// main loops (gemm_kernel.cuh: the 4-wave tiles; gemm8_kernel.cuh: the 8-wave loop, whose waves 4 .. 7 leave a producer workgroup at once).
#pragma once
#include "gemm_epilogue.cuh"

namespace mi355x {

constexpr int LORA_RC = 32;    // ranks per up-projection step (one K step of the epilogue product)
constexpr int LORA_PM = 32;    // rows per LoRA producer workgroup: small blocks = many short workgroups with a deep LDS ring (latency-bound loop)
constexpr int LORA_RMAX = 128;  // largest stacked rank handled inside a launch (control-lora-*-rank128)

// ---- LoRA producer ---------------------------------------------------------------------------------------------------------------
// t[m0 .. m0 + 32)[0 .. R) = x A_g^T for one (column group, 32-row block), R = 32 RI.  A SEPARATE, non-inlined function called at the very
// top of gemm_kernel by the workgroups at the head of a LoRA launch's grid: compiled on its own, it does not touch the register allocation
// or the instruction stream of the tiles' path (inlined, its mere presence cost every tile ~0.8 us: profiles/r03_g_bisect.log).
// The loop is latency-bound (R / BN of a tile's MFMAs on a quarter of its rows), so the workgroup's LDS ring is re-cut into PST <= 8 stages of
// (32 x rows + R weight rows) x 128 B, PST - 1 K blocks in flight.  Own loader: one 16-byte piece of x per thread and K block (plain rows,
// K-blocked rows, or the taps of a convolution gathered from the NHWC image) + RI pieces of the stacked down rows (always K-blocked:
// [K blocks][R][128 B]; LDS row r = rank r).  Wave w multiplies row block w & 1 against the RI rank blocks (w >> 1) RI ...
// LayerNorm folded in -> what is published is t / rstd = (x A'^T - mean sA) + cA / rstd (the tile epilogue's rstd * (acc - mean s) + c then
// scales the up-projected product back: one rounding of t, as in the reference); (mean, M2) of a row = the producer launch's 32-column
// partials Chan-merged in index order.  t is rounded to T, written through to L2 (8-byte agent-scope stores), then the block's flag.
// (Round 4 tried the opposite design -- operands streamed straight into MFMA fragment layout through registers, 2-6 K blocks in flight per
//  wave, no LDS, no barrier, one producer per row block serving all column groups: correct, and 1.3-2x SLOWER per launch (N = K = 1280:
//  27.0 vs 17.5 us; Q|K|V^T 62.6 vs 40.2; step 31.05 vs 25.9 ms, profiles/r04_b_*): beside tiles that keep the CU's vector-memory queue full a
//  producer pays 2-3 us per dependent round trip whatever it asks for, and fragment-shaped loads put 4x the lines through that queue.)
// (Also tried and removed in round 4: TWO producers per 32-row block, one per half of K, the second adding the first's float32 partial before it
//  corrects, rounds and publishes -- for the 64 x 64-tile classes whose producers are the launch's critical path (15.7 us against tiles of 13.9).
//  Correct (58 kernel cases, full-size parity, the two-stream stress test), and slower: N = K = 1280 18.25 vs 16.61 us, FF2 47.45 vs 45.25, step
//  25.638 vs 25.280 ms in the same process (profiles/r04_h_probe_lora_ksplit.log, r04_h_ab_ksplit.log): twice the producer workgroups beside the
//  tiles and a second dependent hand-off cost more than the halved K loop returns.)
// (And: 16-row producers for the rank-32 launches of the 64 x 64 tile -- five ring stages in the 32 KB instead of three K blocks in flight, twice the
//  flags.  Correct on the same cases; per launch indistinguishable (N = K = 1280: 17.40 vs 17.42 us; FF2 46.27 vs 46.23), step 26.367 vs 26.260 ms in
//  the same process (profiles/r04_i_probe_lora_pm16.log, r04_i_ab_pm16.log): ring depth is not what holds these producers back either.)
// `tid_in` / `lds`: the 8-wave loop's workgroups run TWO producers side by side (waves 0-3 and 4-7, each with its own half of the stage buffers, both passing
// the workgroup's barriers together: equal trip counts): thread id within the producer and the producer's ring; defaults = the whole 256-thread workgroup.
template <typename T, bool CONV, int RI, int PST>
__device__ __forceinline__ void lora_producer(const GemmP& p, int q, int tid_in = -1, char* lds = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    char* const smem = lds ? lds : smem_all;
    constexpr int NTHR = 256, PXB = LORA_PM * 128, PSTAGE = PXB + 32 * RI * 128, PD = PST - 1, PL = 1 + RI;
    static_assert(PST >= 2 && PST <= 8, "LoRA producer: 2..8 stages");
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int npb = (p.M + LORA_PM - 1) / LORA_PM;
    const int pgi = q / npb, tm = q - pgi * npb, m0 = tm * LORA_PM;
    const int tag = *p.lora_epoch;
    if (p.lora_dbg & 32) __builtin_amdgcn_s_setprio(3);  // (probing: producers' instructions win the CU's issue arbitration against co-resident tiles)
    const SegP& sp = p.seg[0];  // the LoRAs adapt segment 0 (the conv / Linear itself, not a fused shortcut)
    const int nkb = sp.nkb;
    // ---- this thread's piece of the x tile: row tid >> 3, logical chunk tid & 7 (swizzled source chunk, lane-linear LDS image) ----
    const int row = tid >> 3, pch = tid & 7;
    const int xcoff = (pch ^ swz<128>(row)) << 4;
    const bool xvalid = m0 + row < p.M;
    const int xm = xvalid ? m0 + row : p.M - 1;
    int xb = 0, xoy = 0, xox = 0;
    if constexpr (CONV) {
        const int ohw = p.OH * p.OW;
        xb = xm / ohw;
        const int rem = xm - xb * ohw;
        xoy = rem / p.OW;
        xox = rem - xoy * p.OW;
    }
    const char* xbase = nullptr;
    int64_t xoff = 0, woff = 0;
    const int64_t xstep = CONV ? 128 : (sp.xkb ? (int64_t)p.M * 128 : 128), wstep = (int64_t)p.lora_r * 128;
    int tap = 0, cb = 0;
    auto set_tap = [&]() __attribute__((always_inline)) {
        int dy = tap / sp.ksize, dx = tap - dy * sp.ksize;
        dy -= sp.pad;
        dx -= sp.pad;
        const int iy = xoy * sp.stride + dy, ix = xox * sp.stride + dx;
        const int HH = sp.H << sp.ups_shift, WW = sp.W << sp.ups_shift;
        const bool ok = xvalid && iy >= 0 && iy < HH && ix >= 0 && ix < WW;
        const int sy = iy >> sp.ups_shift, sx = ix >> sp.ups_shift;
        const int64_t pix = ((int64_t)xb * sp.H + sy) * sp.W + sx;
        xbase = ok ? sp.x + pix * sp.ldxb + xcoff : nullptr;
    };
    if constexpr (CONV) set_tap();
    else xbase = sp.x + (sp.xkb ? (int64_t)xm * 128 : (int64_t)xm * sp.ldxb) + xcoff;
    const char* pw[RI];
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int qq = j * NTHR + tid, r = qq >> 3, c = qq & 7;
        pw[j] = p.lora_a[pgi] + (int64_t)r * 128 + ((c ^ swz<128>(r)) << 4);
    }
    int kb = 0;
    auto issue_p = [&](int buf) __attribute__((always_inline)) {
        char* st = smem + buf * PSTAGE;
        const char* src;
        if constexpr (CONV) src = xbase ? xbase + (int64_t)cb * 128 : p.zeros + xcoff;
        else src = xbase + xoff;
        glds16(src, st + wid * 64 * 16);
#pragma unroll
        for (int j = 0; j < RI; ++j) glds16(pw[j] + woff, st + PXB + (j * NTHR + wid * 64) * 16);
        ++kb;
        woff += wstep;
        if constexpr (CONV) {
            if (++cb == sp.cpb) {
                cb = 0;
                ++tap;
                if (kb < nkb) set_tap();
            }
        } else {
            xoff += xstep;
        }
    };
    // ---- LayerNorm folded in: (mean, 1 / rstd) of this lane's row 16 rb + c16.  The producer launch's 32-column partials are requested
    // BEFORE the first stages and merged after their issue: one round trip, overlapped with the stages' (a serial load-merge chain
    // would be ln_parts dependent L2 round trips at the head of every producer).
    const int rb = wid & 1, rg = wid >> 1;
    const int mrow = 16 * rb + c16, m = m0 + mrow;
    float mean = 0.f, inv = 1.f;
    constexpr int MAXP = 48;  // partials held in registers at once (K <= 1536 in one batch)
    f32x2 lst[MAXP];
    const float* sp2 = p.ln_stats ? p.ln_stats + (int64_t)min(m, p.M - 1) * 2 : nullptr;
    const int64_t pstride = (int64_t)p.M * 2;
    if (sp2) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i) lst[i] = i < p.ln_parts ? *reinterpret_cast<const f32x2*>(sp2 + i * pstride) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int s0 = 0; s0 < PD; ++s0)
        if (s0 < nkb) issue_p(s0);
    if (sp2) {
        float m2 = 0.f, cn = 0.f;
#pragma unroll
        for (int i = 0; i < MAXP; ++i)
            if (i < p.ln_parts) stat_merge(cn, mean, m2, 32.f, lst[i][0], lst[i][1]);
        for (int part = MAXP; part < p.ln_parts; ++part) {  // wider rows: the slow way
            const f32x2 s2 = *reinterpret_cast<const f32x2*>(sp2 + part * pstride);
            stat_merge(cn, mean, m2, 32.f, s2[0], s2[1]);
        }
        inv = sqrtf(m2 / cn + p.ln_eps);  // 1 / rstd
    }
    f32x4 ta[RI];
#pragma unroll
    for (int j = 0; j < RI; ++j) ta[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nkb; ++t) {
        if (t + PD <= nkb) wait_vm<(PD - 1) * PL>();  // block t has landed, the PD - 1 younger ones stay in flight
        else wait_vm0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + PD < nkb) issue_p((t + PD) % PST);  // into the buffer block t - 1 was read from (every wave retired those reads above)
        const char* st = smem + (t % PST) * PSTAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const frag_t xf = lds_read_frag(st, tile_off<128>(16 * rb + c16, 4 * kk + g));
#pragma unroll
            for (int j = 0; j < RI; ++j) {
                const frag_t af = lds_read_frag(st + PXB, tile_off<128>((rg * RI + j) * 16 + c16, 4 * kk + g));
                mma_step<T>(ta[j], af, xf);  // D[rank 16 (rg RI + j) + 4 g + r][row 16 rb + c16]
            }
        }
    }
    char* tg = p.lora_t + pgi * p.lora_gs;
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int r0 = (rg * RI + j) * 16 + 4 * g;
        f32x4 v = ta[j];
        if (p.ln_stats) {
            const f32x4 sa = *reinterpret_cast<const f32x4*>(p.lora_ls + pgi * p.lora_r + r0), ca = *reinterpret_cast<const f32x4*>(p.lora_lc + pgi * p.lora_r + r0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean * sa[e]) + ca[e] * inv;
        }
        if (m < p.M) {
            char* dst = tg + ((int64_t)m * p.lora_r + r0) * (int)sizeof(T);
            if constexpr (sizeof(T) == 4) {
                st_agent8(dst, __builtin_bit_cast(uint64_t, f32x2{v[0], v[1]}));
                st_agent8(dst + 8, __builtin_bit_cast(uint64_t, f32x2{v[2], v[3]}));
            } else {
                const bf16x4 b4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                st_agent8(dst, __builtin_bit_cast(uint64_t, b4));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.lora_flags + pgi * npb + tm, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace mi355x
