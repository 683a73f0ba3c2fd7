// mi355x_gemm: LDS-tiled MFMA GEMM / implicit-GEMM convolution for gfx950 with fused epilogues.
//
//   out[M,N] = epi( sum_s X_s[M,K_s] . W_s[N,K_s]^T )        (see include/mi355x_refiners.h for the contract)
//
// Structure (one workgroup = WM x WN waves, BM x BN output tile, K consumed in 128-byte blocks per row):
//   * both operands are K-contiguous, so an LDS tile is `rows x 128 B`; the global->LDS copy is
//     global_load_lds_dwordx4 (16 B per lane, lane-linear LDS image) with the bank-conflict XOR swizzle applied to
//     the per-lane SOURCE chunk; two LDS stages, one barrier per K block (load of block k+1 overlaps MFMA on block k);
//   * MFMA orientation: A operand = weight rows, B operand = activation rows, so a lane ends up holding, for each of
//     its activation rows, 4 consecutive N per 16x16 tile.  The weight tile is loaded with the row permutation
//     R = 16j + 4a + b  <->  n = 4*NT*a + 4j + b, which makes every lane own 4*NT CONSECUTIVE output columns:
//     the epilogue (bias, time-embedding row bias, GEGLU, residual) is fully 16-byte vectorised;
//   * conv mode gathers the activation rows straight from the NHWC image (zero padding comes from a zero page, nearest
//     2x upsampling and stride 2 are address arithmetic), so no im2col buffer, no materialised upsample / concat;
//   * bf16 -> v_mfma_f32_16x16x32_bf16, f32 (parity mode) -> v_mfma_f32_16x16x4_f32; identical LDS image.
#include "common.cuh"
#include "../../include/mi355x_refiners.h"

namespace {

struct SegP {
    const char* x;
    const char* w;
    int64_t ldxb;  // bytes
    int64_t ldwb;  // bytes
    int nkb;       // number of 128-byte K blocks in this segment
    int cpb;       // conv: K blocks per tap (= channels*sizeof(T)/128)
    int ksize, stride, ups_shift, H, W;
    int wkb, xkb;  // operand stored K-blocked: [K block][row][128 B]
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
    int tile_hint;                    // 0 = heuristic, 1..5 = caller's choice
    int out_kb;                       // GEGLU output stored K-blocked ([column block][M rows][128 B]) for the GEMM that consumes it as x
    int krot;                         // debug: rotate each workgroup's K-block order (single-segment, non-split launches)
    // weight prefetch for the NEXT launch: the first pf_blocks workgroups of the grid do no tile work, they touch every 64 bytes of
    // [pf_ptr, pf_ptr + pf_bytes) so that those lines sit in the Infinity Cache when the next kernel asks for them
    const char* pf_ptr[MI355X_MAX_PREFETCH];
    int64_t pf_bytes[MI355X_MAX_PREFETCH];
    int pf_blocks, pf_mode;           // pf_mode: 1 = plain loads, 2 = non-temporal loads (L2 evict-first)
    int pn, hm, hn;  // XCD rasterisation: the 8 XCDs own a pm x pn grid of hm x hn-tile regions
    int vec_ok;
};

// ABL (ablation, probing only): 0 = the real kernel; 1 = no MFMA; 2 = no LDS fragment reads; 3 = no global->LDS loads;
// 4 = the real kernel with the older loader that recomputes every address from the segment descriptor in every iteration
template <typename T, int BM, int BN, int WM, int WN, bool CONV, int NSTAGE, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmP p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
    constexpr int XI = BM * 8 / NTHR, WI = BN * 8 / NTHR;
    constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 LDS stages");
    constexpr int WNE = 16 * NT;  // columns per wave
    static_assert(BM * 8 % NTHR == 0 && BN * 8 % NTHR == 0, "tile/thread mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id();
    const int g = lane >> 4, c16 = lane & 15;
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware rasterisation: workgroup b runs on XCD b % 8 (observed dispatch rule, a speed assumption only); each XCD
    // owns one rectangular region of the tile grid so that its private L2 sees as few distinct operand rows as possible.
    const int pf_first = p.pf_mode >= 3 ? (int)gridDim.x - p.pf_blocks : 0;  // pf_mode 3: prefetchers at the tail of the grid (probe)
    if ((int)blockIdx.x >= pf_first && (int)blockIdx.x < pf_first + p.pf_blocks) {  // prefetch role (see GemmP::pf_ptr): one 4-byte read per 64 bytes, 8 independent loads in flight
        int acc = 0;
        const int64_t stride = (int64_t)p.pf_blocks * NTHR * 64;
        constexpr int U = 8;
#pragma unroll
        for (int sp = 0; sp < MI355X_MAX_PREFETCH; ++sp) {
            const char* base = p.pf_ptr[sp];
            const int64_t bytes = base ? p.pf_bytes[sp] : 0;
            for (int64_t off = ((int64_t)(blockIdx.x - pf_first) * NTHR + tid) * 64; off < bytes; off += stride * U) {
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
    const int bid = p.pf_mode >= 3 ? (int)blockIdx.x : (int)blockIdx.x - p.pf_blocks;
    const int split = p.ksplit > 1 ? bid / p.grid0 : 0;
    const int bx = bid - split * p.grid0;
    int tm, tn;
    if (p.pn > 0) {  // rectangular regions (exact split of the tile grid, grid0 = 8 * hm * hn)
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

    // ---- per-thread loader coordinates (fixed for the whole K loop) ----
    int xm[XI];      // clamped global row (plain) / global row (conv)
    int xcoff[XI];   // logical chunk * 16
    int xb[XI], xoy[XI], xox[XI];
    bool xvalid[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int q = it * NTHR + tid, row = q >> 3, pch = q & 7;
        xcoff[it] = (pch ^ swz<128>(row)) << 4;
        const int m = m0 + row;
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
        int n = n0 + (row - rl) + 4 * NT * a + 4 * j + b;
        wnrow[it] = n < p.N ? n : p.N - 1;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- K-block iteration state ----
    int seg = 0, kb = 0;  // kb = block index inside the current segment
    int total_kb = 0;
    for (int s = 0; s < p.nseg; ++s) total_kb += p.seg[s].nkb;
    // K rotation: workgroups that run concurrently on one XCD start at different K blocks (and wrap around), so that at any
    // instant they read different 128-byte columns of the operand rows instead of all hammering the same L2 channels
    const bool rotate = p.krot && p.nseg == 1 && p.ksplit == 1;
    if (rotate) kb = ((bx >> 3) * p.krot) % total_kb;
    if (p.ksplit > 1) {  // this workgroup's share of the K blocks: [first, first + total_kb)
        const int first = split * p.kb_per_split;
        total_kb = min(p.kb_per_split, total_kb - first);
        kb = first;
        while (seg < p.nseg - 1 && kb >= p.seg[seg].nkb) {
            kb -= p.seg[seg].nkb;
            ++seg;
        }
    }

    auto issue_legacy = [&](int buf) {
        const SegP& sp = p.seg[seg];
        char* xs = smem + buf * STAGE;
        char* ws = xs + XBYTES;
        int dy = 0, dx = 0, cb = kb;
        if constexpr (CONV) {
            const int tap = kb / sp.cpb;
            cb = kb - tap * sp.cpb;
            dy = tap / sp.ksize;
            dx = tap - dy * sp.ksize;
            dy -= sp.pad;
            dx -= sp.pad;
        }
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const char* src;
            if constexpr (CONV) {
                const int iy = xoy[it] * sp.stride + dy, ix = xox[it] * sp.stride + dx;
                const int HH = sp.H << sp.ups_shift, WW = sp.W << sp.ups_shift;
                const bool ok = xvalid[it] && iy >= 0 && iy < HH && ix >= 0 && ix < WW;
                const int sy = iy >> sp.ups_shift, sx = ix >> sp.ups_shift;
                const int64_t pix = ((int64_t)xb[it] * sp.H + sy) * sp.W + sx;
                src = ok ? sp.x + pix * sp.ldxb + (int64_t)cb * 128 + xcoff[it] : p.zeros + xcoff[it];
            } else {
                src = sp.xkb ? sp.x + ((int64_t)kb * p.M + xm[it]) * 128 + xcoff[it] : sp.x + (int64_t)xm[it] * sp.ldxb + (int64_t)kb * 128 + xcoff[it];
            }
            if constexpr (ABL != 3) glds16(src, xs + (it * NTHR + wid * 64) * 16);
            else asm volatile("" ::"v"(src));
        }
#pragma unroll
        for (int it = 0; it < WI; ++it) {
            const char* src = sp.wkb ? sp.w + ((int64_t)kb * p.N + wnrow[it]) * 128 + wcoff[it] : sp.w + (int64_t)wnrow[it] * sp.ldwb + (int64_t)kb * 128 + wcoff[it];
            if constexpr (ABL != 3) glds16(src, ws + (it * NTHR + wid * 64) * 16);
            else asm volatile("" ::"v"(src));
        }
        // advance
        if (++kb == sp.nkb) {
            kb = 0;
            if (!rotate) ++seg;
        }
    };
    // ---- fast loader state (ABL != 4): everything that does not change from one K block to the next is hoisted out of the loop.
    // Per thread: the row base pointers of the current segment (conv: of the current tap) with the swizzled chunk offset folded
    // in; per workgroup: a running byte offset along K.  The per-iteration cost of a load is one 64-bit add; the segment
    // descriptor (a dynamically indexed kernel argument, i.e. scalar loads + waits) is touched only when the segment or tap changes.
    const char* xbase[XI];
    const char* wbase[WI];
    int64_t xstep = 128, wstep = 128;  // bytes from one K block to the next (row-major: 128; K-blocked: rows * 128)
    int64_t xoff = 0, woff = 0;        // running offsets inside the current segment
    int cur_nkb = 0, cur_cpb = 1, tap = 0, cb = 0;
    auto set_tap = [&](const SegP& sp) {  // conv: per-thread pixel pointers of tap `tap` (zero page for padding / out-of-tile rows)
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
    auto enter = [&](int s, int kb0) {  // make segment s current, positioned at its K block kb0
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
    if constexpr (ABL != 4) enter(seg, kb);

    auto issue = [&](int buf) {
        if constexpr (ABL == 4) {
            issue_legacy(buf);
        } else {
            char* xs = smem + buf * STAGE;
            char* ws = xs + XBYTES;
#pragma unroll
            for (int it = 0; it < XI; ++it) {
                const char* src;
                if constexpr (CONV) src = xbase[it] ? xbase[it] + (int64_t)cb * 128 : p.zeros + xcoff[it];
                else src = xbase[it] + xoff;
                if constexpr (ABL != 3) glds16(src, xs + (it * NTHR + wid * 64) * 16);
                else asm volatile("" ::"v"(src));
            }
#pragma unroll
            for (int it = 0; it < WI; ++it) {
                const char* src = wbase[it] + woff;
                if constexpr (ABL != 3) glds16(src, ws + (it * NTHR + wid * 64) * 16);
                else asm volatile("" ::"v"(src));
            }
            // advance along K; cross into the next tap / segment when this one is exhausted
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
                if (!rotate) ++seg;
                if (seg < p.nseg) enter(seg, 0);
            }
        }
    };
    auto compute = [&](int buf) {
        const char* xs = smem + buf * STAGE;
        const char* ws = xs + XBYTES;
        // all 2 x (MT + NT) fragment reads of the K block are issued before the first MFMA, so that the LDS latency of the
        // second half overlaps the matrix work of the first (the compiler then waits with counted lgkmcnt, not lgkmcnt(0) twice)
        frag_t xf[2][MT], wf[2][NT];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (ABL != 2) xf[kk][i] = lds_read_frag(xs, tile_off<128>(wm * 16 * MT + 16 * i + c16, 4 * kk + g));
                else xf[kk][i] = frag_t{i + buf, kk, lane, 1};
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (ABL != 2) wf[kk][j] = lds_read_frag(ws, tile_off<128>(wn * WNE + 16 * j + c16, 4 * kk + g));
                else wf[kk][j] = frag_t{j, kk + buf, lane, 2};
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (ABL != 1) mma_step<T>(acc[i][j], wf[kk][j], xf[kk][i]);
                    else asm volatile("" ::"v"(wf[kk][j]), "v"(xf[kk][i]));
                }
    };

    // ---- software pipeline: NSTAGE LDS buffers, D = NSTAGE - 1 K blocks in flight -------------------------------------
    // per iteration: counted vmcnt (block t has landed, the D-1 younger ones stay in flight) -> raw barrier (no vmcnt(0)
    // drain, guide section 5 "pipelining across barriers") -> issue block t+D into the buffer block t-1 was computed from
    // -> MFMA on block t.  One barrier per K block.
    constexpr int D = NSTAGE - 1;
    constexpr int LPS = XI + WI;  // global_load_lds instructions per thread per stage
#pragma unroll
    for (int s0 = 0; s0 < D; ++s0)
        if (s0 < total_kb) issue(s0);
    for (int t = 0; t < total_kb; ++t) {
        if (t + D <= total_kb) wait_vm<(D - 1) * LPS>();
        else wait_vm0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + D < total_kb) issue((t + D) % NSTAGE);
        compute(t % NSTAGE);
    }

    // ---- epilogue: every lane owns RUN = 4*NT consecutive columns of MT rows ----
    constexpr int RUN = 4 * NT;
    constexpr int EPC = DT<T>::EPC;
    T* out = reinterpret_cast<T*>(p.out);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowbias = reinterpret_cast<const T*>(p.rowbias);
    const T* res = reinterpret_cast<const T*>(p.res);
    const int nl = wn * WNE + RUN * g;
    const int n = n0 + nl;
    const bool full = p.vec_ok && (n + RUN <= p.N);
    if (p.ksplit > 1) {  // split-K: raw float32 partial sums; bias / residual / conversion happen in splitk_reduce_kernel
        float* part = p.partial + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 16 * MT + 16 * i + c16;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nn = n + 4 * j;
                if (nn + 4 <= p.N) *reinterpret_cast<f32x4*>(part + (int64_t)m * p.N + nn) = acc[i][j];
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nn + r < p.N) part[(int64_t)m * p.N + nn + r] = acc[i][j][r];
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * 16 * MT + 16 * i + c16;
        if (m >= p.M) continue;
        float v[RUN];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
        if (full) {
            if (bias) {
#pragma unroll
                for (int c = 0; c < RUN / EPC; ++c) {
                    Vec16<T> bv = load16<T>(bias + n + c * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[c * EPC + e] += bv.get(e);
                }
            }
            if (rowbias) {
                const T* rb = rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias + n;
#pragma unroll
                for (int c = 0; c < RUN / EPC; ++c) {
                    Vec16<T> bv = load16<T>(rb + c * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[c * EPC + e] += bv.get(e);
                }
            }
            if (p.gelu) {
#pragma unroll
                for (int e = 0; e < RUN; ++e) v[e] = p.gelu == 1 ? gelu_exact(v[e]) : quick_gelu(v[e]);
            }
            if (p.geglu) {
                if constexpr (NT == 4) {
                    constexpr int HR = RUN / 2;
                    const int no = (n0 + wn * WNE) / 2 + HR * g;
                    float o[HR];
#pragma unroll
                    for (int e = 0; e < HR; ++e) o[e] = v[e] * gelu_exact(v[HR + e]);
                    if (res) {
                        const T* rp = res + (int64_t)m * p.ldres + no;
#pragma unroll
                        for (int c = 0; c < HR / EPC; ++c) {
                            Vec16<T> rv = load16<T>(rp + c * EPC);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) o[c * EPC + e] += rv.get(e);
                        }
                    }
                    constexpr int BKE = 128 / (int)sizeof(T);  // elements per 128-byte K block of the consumer
                    T* op = p.out_kb ? out + ((int64_t)(no / BKE) * p.M + m) * BKE + no % BKE : out + (int64_t)m * p.ldo + no;
#pragma unroll
                    for (int c = 0; c < HR / EPC; ++c) {
                        Vec16<T> ov;
#pragma unroll
                        for (int e = 0; e < EPC; ++e) ov.set(e, o[c * EPC + e]);
                        store16<T>(op + c * EPC, ov);
                    }
                }
            } else {
                if (res) {
                    const T* rp = res + (int64_t)m * p.ldres + n;
#pragma unroll
                    for (int c = 0; c < RUN / EPC; ++c) {
                        Vec16<T> rv = load16<T>(rp + c * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[c * EPC + e] += rv.get(e);
                    }
                }
                T* op = out + (int64_t)m * p.ldo + n;
#pragma unroll
                for (int c = 0; c < RUN / EPC; ++c) {
                    Vec16<T> ov;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) ov.set(e, v[c * EPC + e]);
                    store16<T>(op + c * EPC, ov);
                }
            }
        } else {
            // guarded scalar path (N edge tiles, unaligned outputs); geglu is never routed here (host checks)
#pragma unroll
            for (int e = 0; e < RUN; ++e) {
                const int nn = n + e;
                if (nn < p.N) {
                    float val = v[e];
                    if (bias) val += to_f32(bias[nn]);
                    if (rowbias) val += to_f32(rowbias[(int64_t)(m / p.rows_per_group) * p.ld_rowbias + nn]);
                    if (p.gelu) val = p.gelu == 1 ? gelu_exact(val) : quick_gelu(val);
                    if (res) val += to_f32(res[(int64_t)m * p.ldres + nn]);
                    out[(int64_t)m * p.ldo + nn] = from_f32<T>(val);
                }
            }
        }
    }
}

// out[m][n] = dtype( sum_s partial[s][m][n] (fixed order) + bias[n] + rowbias[m / rpg][n] + res[m][n] ): 4 columns per thread
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmP p) {
    const int nq = (p.N + 3) / 4;
    const int64_t total = (int64_t)p.M * nq;
    T* out = reinterpret_cast<T*>(p.out);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowbias = reinterpret_cast<const T*>(p.rowbias);
    const T* res = reinterpret_cast<const T*>(p.res);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int m = (int)(q / nq);
        const int n = (int)(q - (int64_t)m * nq) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full4 = n + 4 <= p.N;
        for (int s = 0; s < p.ksplit; ++s) {
            const float* pp = p.partial + ((int64_t)s * p.M + m) * p.N + n;
            if (full4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(pp);
                v[0] += t[0], v[1] += t[1], v[2] += t[2], v[3] += t[3];
            } else {
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] += pp[r];
            }
        }
        for (int r = 0; r < 4; ++r) {
            const int nn = n + r;
            if (nn >= p.N) break;
            float val = v[r];
            if (bias) val += to_f32(bias[nn]);
            if (rowbias) val += to_f32(rowbias[(int64_t)(m / p.rows_per_group) * p.ld_rowbias + nn]);
            if (p.gelu) val = p.gelu == 1 ? gelu_exact(val) : quick_gelu(val);
            if (res) val += to_f32(res[(int64_t)m * p.ldres + nn]);
            out[(int64_t)m * p.ldo + nn] = from_f32<T>(val);
        }
    }
}

int g_pf_blocks = 64;  // default number of prefetch workgroups when the caller gives spans but no count (0 = prefetch off)
int g_pf_mode = 1;    // 1 = plain loads, 2 = non-temporal

template <typename T, int BM, int BN, int WM, int WN, bool CONV, int NSTAGE, int ABL = 0>
int launch_cfg(const GemmP& p, hipStream_t stream) {
    constexpr int LDS = NSTAGE * (BM + BN) * 128;
    auto kfn = gemm_kernel<T, BM, BN, WM, WN, CONV, NSTAGE, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    GemmP q = p;
    q.tiles_n = (p.N + BN - 1) / BN;
    q.tiles_m = (p.M + BM - 1) / BM;
    // How the 8 XCDs (private L2 each) share the tile grid.  Bytes pulled into the L2s ~ nx * |X| + nw * |W| where nx / nw =
    // number of XCDs that touch each activation row / weight row; |X|, |W| in K-elements per row (a 3x3 conv reads every
    // activation row through 9 taps but it is ONE row in L2).  Candidates: exact pm x pn rectangles, or balanced contiguous
    // chunks of the row-major (nx = 1, nw = 8) / column-major (nx = 8, nw = 1) order.
    double kx = 0, kw = 0;
    for (int sgi = 0; sgi < p.nseg; ++sgi) {
        const SegP& sg = p.seg[sgi];
        kw += sg.nkb;
        kx += CONV ? (double)sg.nkb / (sg.ksize * sg.ksize) : (double)sg.nkb;
    }
    const double bx_ = (double)p.M * kx, bw_ = (double)p.N * kw;
    double best = bx_ + 8.0 * bw_;  // row-major chunks
    q.pn = 0;
    q.hm = q.hn = 0;
    if (8.0 * bx_ + bw_ < best) {
        best = 8.0 * bx_ + bw_;
        q.pn = -1;
    }
    for (int pm = 2; pm <= 4; pm *= 2) {
        const int pn = 8 / pm;
        if (q.tiles_m % pm || q.tiles_n % pn) continue;
        const double cost = pn * bx_ + pm * bw_;
        if (cost < best) {
            best = cost;
            q.pn = pn;
            q.hm = q.tiles_m / pm;
            q.hn = q.tiles_n / pn;
        }
    }
    q.grid0 = q.tiles_m * q.tiles_n;
    bool any_pf = false;
    for (int i = 0; i < MI355X_MAX_PREFETCH; ++i) any_pf = any_pf || (q.pf_ptr[i] && q.pf_bytes[i] > 0);
    if (!any_pf || g_pf_blocks == 0) q.pf_blocks = 0;
    else if (q.pf_blocks <= 0) q.pf_blocks = g_pf_blocks;
    q.pf_blocks = (q.pf_blocks + 7) / 8 * 8;  // a multiple of 8: compute block b still lands on XCD b % 8
    q.pf_mode = g_pf_mode;
    const int grid = q.pf_blocks + q.grid0 * (q.ksplit > 1 ? q.ksplit : 1);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WM * WN * 64), LDS, stream, q);
    if (q.ksplit > 1) {
        const int64_t work = (int64_t)q.M * ((q.N + 3) / 4);
        int64_t rb = (work + 255) / 256;
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((int)rb), dim3(256), 0, stream, q);
    }
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

int g_tile = 0;    // 0 = heuristic, 1..4 = force a tile configuration (probing / A-B runs)
int g_stages = 0;  // 0 = heuristic, 2..4 = force the LDS pipeline depth
int g_krot = 0;    // 0 = off, n = workgroup i of an XCD starts its K loop at block (i * n) % nkb

// Tile configurations (all 4 waves, 2 x 2):  1: 128x128   2: 128x64   3: 64x128   4: 64x64
// The UNet's GEMMs are small for a 256-CU chip (2048x1280 outputs = 160 tiles of 128x128), so the choice is driven by
// how many workgroups a configuration yields: big tiles reuse operands better, small tiles fill the machine.
int g_alt = 0;  // probing: alternative tile / stage heuristics
int g_legacy = 0;  // probing: 1 = the pre-hoisting loader (per-iteration address arithmetic, ABL = 4) for A/B runs
inline int pick_stages_default(const GemmP&, int);
inline int pick_tile(const GemmP& p, bool conv) {
    // measured on MI355X over the UNet's shapes (tools/probe_gemm.py, profiles/r01_b_probe_gemm_tiles.log)
    if (g_tile >= 1 && g_tile <= 5) return g_tile;
    if (p.tile_hint >= 1 && p.tile_hint <= 5) return p.tile_hint;
    const int64_t b128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (g_alt >= 1 && g_alt <= 3 && !conv && !p.geglu && b128 <= 256) {  // probing: the N = 1280 class of the SDXL step
        int total_kb = 0;
        for (int s = 0; s < p.nseg; ++s) total_kb += p.seg[s].nkb;
        if (g_alt != 3 || total_kb >= 64) return 3;
    }
    if (conv) return 3;  // 64 x 128 wins for every conv shape of the UNet (r01_b probe: 339 / 540 / 570 TF at 32^2 / 64^2 / 128^2)
    if (p.geglu) return 1;
    if (b128 <= 256) return 4;
    if (b128 < 1000) return 2;
    return 1;
}
inline int pick_stages(const GemmP& p, int tile) {
    if (g_stages == 0 && (g_alt == 1 || g_alt == 3) && tile == 3 && !p.geglu && p.seg[0].ksize == 1 && p.seg[0].stride == 1) {
        const int64_t b128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
        if (b128 <= 256) return 3;
    }
    return pick_stages_default(p, tile);
}
inline int pick_stages_default(const GemmP&, int) {
    // two LDS stages everywhere: deeper pipelines cost a resident workgroup per CU (LDS), and on these short-K GEMMs
    // co-resident workgroups hide latency better than prefetch depth does (same probe).
    if (g_stages >= 2 && g_stages <= 4) return g_stages;
    return 2;
}

template <typename T, int BM, int BN, bool CONV>
int launch_stages(const GemmP& p, int stages, hipStream_t stream) {
    switch (stages) {
        case 2: return g_legacy ? launch_cfg<T, BM, BN, 2, 2, CONV, 2, 4>(p, stream) : launch_cfg<T, BM, BN, 2, 2, CONV, 2>(p, stream);
        case 4: return launch_cfg<T, BM, BN, 2, 2, CONV, 4>(p, stream);
        default: return launch_cfg<T, BM, BN, 2, 2, CONV, 3>(p, stream);
    }
}
// tile 5: 256 x 128, 8 waves (4 x 2) of 64 x 64: 25 % fewer operand bytes per FLOP through the L1 -> LDS path than 128 x 128
template <typename T, bool CONV>
int launch_big(const GemmP& p, int stages, hipStream_t stream) {
    if (stages == 3) return launch_cfg<T, 256, 128, 4, 2, CONV, 3>(p, stream);
    return launch_cfg<T, 256, 128, 4, 2, CONV, 2>(p, stream);
}

template <typename T, bool CONV>
int launch_tile(const GemmP& p, hipStream_t stream) {
    int tile = pick_tile(p, CONV);
    if (p.geglu && (tile == 2 || tile == 4)) tile = 3;  // the GEGLU epilogue needs 64 packed columns per wave
    const int st = pick_stages(p, tile);
    switch (tile) {
        case 1: return launch_stages<T, 128, 128, CONV>(p, st, stream);
        case 2: return launch_stages<T, 128, 64, CONV>(p, st, stream);
        case 3: return launch_stages<T, 64, 128, CONV>(p, st, stream);
        case 5: return launch_big<T, CONV>(p, st, stream);
        default: return launch_stages<T, 64, 64, CONV>(p, st, stream);
    }
}

int g_ablate = 0;

template <typename T>
int launch_t(const GemmP& p, bool conv, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) {
        if (g_ablate && !conv) {  // probing only: bf16, plain GEMM, 2 stages; tile 1 (128x128) or 4 (64x64)
            const bool small = g_tile == 4;
            switch (g_ablate) {
                case 1: return small ? launch_cfg<T, 64, 64, 2, 2, false, 2, 1>(p, stream) : launch_cfg<T, 128, 128, 2, 2, false, 2, 1>(p, stream);
                case 2: return small ? launch_cfg<T, 64, 64, 2, 2, false, 2, 2>(p, stream) : launch_cfg<T, 128, 128, 2, 2, false, 2, 2>(p, stream);
                default: return small ? launch_cfg<T, 64, 64, 2, 2, false, 2, 3>(p, stream) : launch_cfg<T, 128, 128, 2, 2, false, 2, 3>(p, stream);
            }
        }
    }
    if (conv) return launch_tile<T, true>(p, stream);
    return launch_tile<T, false>(p, stream);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int mi355x_set_option(const char* name, int value);
extern "C" int mi355x_set_option(const char* name, int value) {
    // debugging / A-B switches; not part of the stable contract
    if (name && name[0] == 'g') return MI355X_OK;  // "glds": the GEMM has a single (global_load_lds) loader now
    if (name && name[0] == 'a') {  // "ablate"
        g_ablate = value;
        return MI355X_OK;
    }
    if (name && name[0] == 's') {  // "stages"
        g_stages = value;
        return MI355X_OK;
    }
    if (name && name[0] == 'p') {  // "pfblocks" / "pfmode"
        if (name[2] == 'b') g_pf_blocks = value < 0 ? 0 : (value + 7) / 8 * 8;
        else g_pf_mode = value;
        return MI355X_OK;
    }
    if (name && name[0] == 'l') {  // "legacy"
        g_legacy = value;
        return MI355X_OK;
    }
    if (name && name[0] == 'h') {  // "heur"
        g_alt = value;
        return MI355X_OK;
    }
    if (name && name[0] == 'k') {  // "krot"
        g_krot = value;
        return MI355X_OK;
    }
    if (name && name[0] == 't') {  // "tile"
        g_tile = value;
        return MI355X_OK;
    }
    return MI355X_EARG;
}

extern "C" int mi355x_gemm(const mi355x_gemm_args* a, void* stream) {
    if (!a || !a->out) return MI355X_EARG;
    if (a->dtype != MI355X_F32 && a->dtype != MI355X_BF16) return MI355X_EDTYPE;
    if (a->M <= 0 || a->N <= 0 || a->nseg < 1 || a->nseg > MI355X_MAX_SEG) return MI355X_ESHAPE;
    const int es = a->dtype == MI355X_F32 ? 4 : 2;
    const int bke = 128 / es;  // elements per K block
    GemmP p{};
    p.M = a->M;
    p.N = a->N;
    p.nseg = a->nseg;
    p.OH = a->OH;
    p.OW = a->OW;
    if (a->conv) {
        if (!a->zeros || a->B <= 0 || a->OH <= 0 || a->OW <= 0 || (int64_t)a->B * a->OH * a->OW != a->M) return MI355X_ESHAPE;
    }
    for (int s = 0; s < a->nseg; ++s) {
        const mi355x_gemm_seg& g = a->seg[s];
        if (!g.x || !g.w || g.k <= 0 || g.k % bke) return MI355X_ESHAPE;
        const bool wkb = g.kblocked & 1, xkb = (g.kblocked & 2) != 0;
        if (xkb && a->conv) return MI355X_ESHAPE;  // the conv loader gathers taps from an image: only its weights can be K-blocked
        if (!aligned16(g.x) || !aligned16(g.w) || (!xkb && (g.ldx * es) % 16) || (!wkb && (g.ldw * es) % 16)) return MI355X_ESHAPE;
        SegP& d = p.seg[s];
        d.wkb = wkb ? 1 : 0;
        d.xkb = xkb ? 1 : 0;
        d.x = static_cast<const char*>(g.x);
        d.w = static_cast<const char*>(g.w);
        d.ldxb = g.ldx * es;
        d.ldwb = g.ldw * es;
        if (a->conv) {
            if ((g.ksize != 1 && g.ksize != 3) || (g.stride != 1 && g.stride != 2) || (g.ups != 1 && g.ups != 2)) return MI355X_ESHAPE;
            if (g.H <= 0 || g.W <= 0) return MI355X_ESHAPE;
            d.cpb = g.k / bke;
            d.nkb = g.ksize * g.ksize * d.cpb;
            d.ksize = g.ksize;
            d.pad = g.asym ? 0 : g.ksize / 2;
            d.stride = g.stride;
            d.ups_shift = g.ups == 2 ? 1 : 0;
            d.H = g.H;
            d.W = g.W;
        } else {
            d.cpb = 1;
            d.nkb = g.k / bke;
            d.ksize = 1;
            d.pad = 0;
            d.stride = 1;
            d.ups_shift = 0;
        }
    }
    p.out = static_cast<char*>(a->out);
    p.ldo = a->ldo;
    p.bias = static_cast<const char*>(a->bias);
    p.rowbias = static_cast<const char*>(a->rowbias);
    p.ld_rowbias = a->ld_rowbias;
    p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1;
    p.geglu = a->geglu == 1 ? 1 : 0;
    p.gelu = a->geglu == 2 ? 1 : (a->geglu == 3 ? 2 : 0);
    p.res = static_cast<const char*>(a->res);
    p.ldres = a->ldres;
    p.zeros = static_cast<const char*>(a->zeros);
    bool vec = aligned16(a->out) && (a->ldo * es) % 16 == 0;
    if (a->bias) vec = vec && aligned16(a->bias);
    if (a->rowbias) vec = vec && aligned16(a->rowbias) && (a->ld_rowbias * es) % 16 == 0;
    if (a->res) vec = vec && aligned16(a->res) && (a->ldres * es) % 16 == 0;
    p.vec_ok = vec ? 1 : 0;
    if (p.geglu && (!vec || a->N % 64)) return MI355X_ESHAPE;
    p.out_kb = a->out_kblocked ? 1 : 0;
    if (p.out_kb && (!p.geglu || a->res || a->N % 256 || a->ksplit > 1)) return MI355X_ESHAPE;  // only the fused-GEGLU store path writes it
    p.tile_hint = a->tile;
    p.krot = g_krot;
    for (int i = 0; i < MI355X_MAX_PREFETCH; ++i) {
        p.pf_ptr[i] = static_cast<const char*>(a->prefetch[i]);
        p.pf_bytes[i] = a->prefetch[i] ? a->prefetch_bytes[i] : 0;
    }
    p.pf_blocks = a->prefetch_blocks;
    p.ksplit = 1;
    p.partial = nullptr;
    if (a->ksplit > 1) {
        int total_kb = 0;
        for (int s = 0; s < a->nseg; ++s) total_kb += p.seg[s].nkb;
        if (p.geglu || !a->ws || (reinterpret_cast<uintptr_t>(a->ws) & 15)) return MI355X_EARG;
        const int kbps = (total_kb + a->ksplit - 1) / a->ksplit;
        const int ks = (total_kb + kbps - 1) / kbps;  // no empty split
        if ((int64_t)ks * a->M * a->N * 4 > a->ws_bytes) return MI355X_EARG;
        if (ks > 1) {
            p.ksplit = ks;
            p.kb_per_split = kbps;
            p.partial = static_cast<float*>(a->ws);
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->dtype == MI355X_F32) return launch_t<float>(p, a->conv != 0, st);
    return launch_t<bf16_t>(p, a->conv != 0, st);
}
