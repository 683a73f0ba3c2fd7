// gemm_epilogue.cuh: the pieces of a GEMM workgroup that do not depend on its main loop -- the LayerNorm consumer's row statistics (computed
// while the first stages are in flight) and the tile epilogue (bias, time-embedding row bias, GEGLU / GELU, residual, folded-LayerNorm correction,
// row statistics for the next LayerNorm, column statistics for the next GroupNorm, float32 output, split-K partials, transposed column groups).
// Both main loops (gemm_kernel.cuh, gemm8_kernel.cuh) hand over the same thing: per wave an accumulator block acc[MT][NT] of 16 x 16 MMA tiles in the
// "A = weight rows (permuted), B = activation rows" orientation, so that a lane owns 4 NT consecutive output columns of MT rows (or, for a
// transposed tile, 4 MT consecutive rows of NT columns).
#pragma once
#include "gemm_params.cuh"

// The optional parts of the row-major tile epilogue, as bits of the instance `rows<F>` that a launch runs (tile_epilogue, FAST): a part whose bit is
// missing from F is compiled out of that instance, so a launch's rows run straight-line code for exactly its kind of epilogue.
enum : unsigned { EPI_LN = 1, EPI_RB = 2, EPI_GELU = 4, EPI_GEGLU = 8, EPI_RES = 16, EPI_F32 = 32, EPI_ST = 64, EPI_CS = 128, EPI_ALL = 255 };
#define EPIF(bit, x) (((F) & (bit)) != 0 && (x))

namespace mi355x {

// (mean, rstd) of the tile's BM rows -> rowstat[BM][2] in LDS, from the producer launch's 32-column (mean, M2) partials: TPR threads per row,
// merged in a fixed order.  Called behind the prologue's LDS-DMA issue so that its latency overlaps theirs.
template <int BM, int NTHR_ALL>
MI_DEV void ln_rowstat(const GemmP& p, int m0, int tid_all, float* rowstat) {
    // LayerNorm consumer: (mean, rstd) of the tile's BM rows from the producer's 32-column partials, TPR threads per row,
    // merged in a fixed order.  Placed behind the prologue's LDS-DMA issue so that its latency overlaps theirs (the compiler
    // drains vmcnt before the first use of an ordinary load anyway; the first loop iteration then finds its stage landed).
    constexpr int TPR = NTHR_ALL / BM;
    static_assert(TPR >= 1 && (TPR & (TPR - 1)) == 0 && TPR <= 8, "threads per row");
    const int row = tid_all / TPR, sub = tid_all % TPR;
    const int m = min(m0 + row, p.M - 1);
    float cn = 0.f, mean = 0.f, m2 = 0.f;
    // all of this thread's partials are loaded BEFORE the first merge (independent loads, one L2 round trip instead of
    // ln_parts / TPR serialized ones: 20 dependent trips at the head of every consumer cost more than the LayerNorm launch saved)
    constexpr int MAXP = 24;
    const float* sp = p.ln_stats + (int64_t)m * 2;
    const int64_t pstride = (int64_t)p.M * 2;
    f32x2 st[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int part = sub + i * TPR;
        st[i] = part < p.ln_parts ? *reinterpret_cast<const f32x2*>(sp + part * pstride) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
        if (sub + i * TPR < p.ln_parts) {  // Chan's update with equal counts of 32: n = 32 i so far, f = 32 / (32 (i + 1)) is a constant
            const float d = st[i][0] - mean, f = 1.0f / (float)(i + 1);
            mean += d * f;
            m2 += st[i][1] + d * d * (32.0f * i) * f;
            cn = 32.0f * (i + 1);
        }
    for (int part = sub + MAXP * TPR; part < p.ln_parts; part += TPR) {  // wider than MAXP * TPR * 32 columns: the slow way
        const f32x2 s2 = *reinterpret_cast<const f32x2*>(sp + part * pstride);
        stat_merge(cn, mean, m2, 32.f, s2[0], s2[1]);
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
        const float nb = __shfl_xor(cn, o), mb = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
        // both partners must combine in the SAME order to end up with identical bits: lower sub-index first
        if (sub & o) {
            float n2 = nb, me2 = mb, q2 = qb;
            stat_merge(n2, me2, q2, cn, mean, m2);
            cn = n2, mean = me2, m2 = q2;
        } else {
            stat_merge(cn, mean, m2, nb, mb, qb);
        }
    }
    if (sub == 0) {
        rowstat[2 * row] = mean;
        rowstat[2 * row + 1] = rsqrtf(m2 / cn + p.ln_eps);
    }
}

// The tile epilogue.  wm / wn: this wave's position in the workgroup's wave grid; m0 / n0: the tile's origin; tr: transposed tile (workgroup-uniform);
// split: this workgroup's split-K index.
// TR_ONLY: instantiate the transposed-tile path alone (the caller passes tr = true; block shapes the row-major path has no code for).
// FAST: dispatch the row loop to the instance compiled for this launch's kind of epilogue (see EPIF).
template <typename T, int MT, int NT, int BM, bool CONV, bool TR_ONLY = false, bool FAST = false>
// colvec (LDS, or null): the tile's per-column vectors staged by the caller, [0, BN): the bias as float32 or the folded LayerNorm's s, [BN, 2 BN): its c (BN = 256);
// a workgroup that owns its CU alone reads them from there: a global load issued behind a row's stores waits for the stores (vmcnt is in order and counts them).
MI_DEV void tile_epilogue(const GemmP& p, f32x4 (&acc)[MT][NT], const float* rowstat, int m0, int n0, int wm, int wn, int lane, bool tr, int split, const float* colvec = nullptr) {
    constexpr int WME = 16 * MT, WNE = 16 * NT;
    const int g = lane >> 4, c16 = lane & 15;
    // ---- epilogue ----
    constexpr int EPC = DT<T>::EPC;
    T* out = reinterpret_cast<T*>(p.out);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowbias = reinterpret_cast<const T*>(p.rowbias);
    const T* res = reinterpret_cast<const T*>(p.res);

    if constexpr (!CONV) {
        if (tr) {
            // transposed tile: every lane owns RUN_T = 4*MT consecutive ROWS m of NT columns n = n0 + wn*WNE + 16 j + c16
            constexpr int RUN_T = 4 * MT;
            T* out_t = reinterpret_cast<T*>(p.out_t);
            const int mb = m0 + wm * WME + RUN_T * g;
            const bool fullm = p.vec_ok && mb + RUN_T <= p.M;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * WNE + 16 * j + c16;
                if (n >= p.N) continue;
                float v[RUN_T];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r];
                if (p.ln_stats) {
                    float s, c;  // (two separate loads, not one load through a selected pointer: that would be a FLAT load, which waits on vmcnt)
                    if (colvec) s = colvec[n - n0], c = colvec[256 + n - n0];
                    else s = p.ln_s[n], c = p.ln_c[n];
#pragma unroll
                    for (int e = 0; e < RUN_T; ++e) {
                        const int row = min(wm * WME + RUN_T * g + e, BM - 1);
                        v[e] = rowstat[2 * row + 1] * (v[e] - rowstat[2 * row] * s) + c;
                    }
                } else if (bias) {
                    float b;
                    if (colvec) b = colvec[n - n0];
                    else b = to_f32(bias[n]);
#pragma unroll
                    for (int e = 0; e < RUN_T; ++e) v[e] += b;
                }
                T* op = out_t + (int64_t)(n - p.nt_begin) * p.ldt + mb;
                if (fullm) {
#pragma unroll
                    for (int c = 0; c < RUN_T / EPC; ++c) {
                        Vec16<T> ov;
#pragma unroll
                        for (int e = 0; e < EPC; ++e) ov.set(e, v[c * EPC + e]);
                        store16<T>(op + c * EPC, ov);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < RUN_T; ++e)
                        if (mb + e < p.M) op[e] = from_f32<T>(v[e]);
                }
            }
            return;
        }
    }

    if constexpr (!TR_ONLY) {
    // every lane owns RUN = 4*NT consecutive columns of MT rows
    constexpr int RUN = 4 * NT;
    const int nl = wn * WNE + RUN * g;
    const int n = n0 + nl;
    const bool full = p.vec_ok && (n + RUN <= p.N);
    if (p.ksplit > 1) {  // split-K: raw float32 partial sums; bias / residual / conversion happen in splitk_reduce_kernel
        float* part = p.partial + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * WME + 16 * i + c16;
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
    auto rows = [&](auto fc) __attribute__((always_inline)) {
    constexpr unsigned F = decltype(fc)::value;
    float cs_a[RUN], cs_b[RUN];  // GemmP::colstats: the even row of the current 32-row block
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#if defined(MI355X_G8_ABL) && (MI355X_G8_ABL & 4)
        if ((int)blockIdx.x == p.pf_blocks && threadIdx.x == 0 && p.sk_ws) reinterpret_cast<uint64_t*>(p.sk_ws)[128 + i] = wall_clock64();  // (probing build: row stamps)
#endif
        const int mrow = wm * WME + 16 * i + c16;
        const int m = m0 + mrow;
        // rows beyond M keep going through the arithmetic when statistics are produced (the shuffles below need all lanes);
        // their stores are suppressed
        const bool mok = m < p.M;
        if (!mok && !EPIF(EPI_ST, p.stats_out) && !EPIF(EPI_CS, p.colstats)) continue;
        float v[RUN];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
        if (full) {
            if (EPIF(EPI_LN, p.ln_stats)) {  // y = rstd * (acc - mean * s[n]) + c[n]   (c carries the Linear's bias)
                const float mean = rowstat[2 * mrow], rstd = rowstat[2 * mrow + 1];
#pragma unroll
                for (int c = 0; c < RUN / 4; ++c) {
                    f32x4 sv, cv;
                    if (colvec) sv = *reinterpret_cast<const f32x4*>(colvec + nl + 4 * c), cv = *reinterpret_cast<const f32x4*>(colvec + 256 + nl + 4 * c);
                    else sv = *reinterpret_cast<const f32x4*>(p.ln_s + n + 4 * c), cv = *reinterpret_cast<const f32x4*>(p.ln_c + n + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * c + e] = rstd * (v[4 * c + e] - mean * sv[e]) + cv[e];
                }
            } else if (bias) {
                if (colvec) {
#pragma unroll
                    for (int c = 0; c < RUN / 4; ++c) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(colvec + nl + 4 * c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * c + e] += bv[e];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < RUN / EPC; ++c) {
                        Vec16<T> bv = load16<T>(bias + n + c * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[c * EPC + e] += bv.get(e);
                    }
                }
            }
            if (EPIF(EPI_RB, rowbias) && mok) {
                const T* rb = rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias + n;
#pragma unroll
                for (int c = 0; c < RUN / EPC; ++c) {
                    Vec16<T> bv = load16<T>(rb + c * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[c * EPC + e] += bv.get(e);
                }
            }
            if (EPIF(EPI_GELU, p.gelu)) {
#pragma unroll
                for (int e = 0; e < RUN; ++e) v[e] = p.gelu == 1 ? gelu_exact(v[e]) : quick_gelu(v[e]);
            }
            if (EPIF(EPI_GEGLU, p.geglu)) {
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
                if (EPIF(EPI_RES, res) && mok) {
                    const T* rp = res + (int64_t)m * p.ldres + n;
#pragma unroll
                    for (int c = 0; c < RUN / EPC; ++c) {
                        Vec16<T> rv = load16<T>(rp + c * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[c * EPC + e] += rv.get(e);
                    }
                }
                if (EPIF(EPI_F32, p.out_f32)) {
                    if (mok) {
                        float* of = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n;
#pragma unroll
                        for (int c = 0; c < RUN / 4; ++c) *reinterpret_cast<f32x4*>(of + 4 * c) = f32x4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
                    }
                    continue;
                }
                T* op = out + (int64_t)m * p.ldo + n;
                float rs = 0.f;  // sum of the values AS STORED (rounded to T): the next LayerNorm normalises the stored tensor
#pragma unroll
                for (int c = 0; c < RUN / EPC; ++c) {
                    Vec16<T> ov;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) ov.set(e, v[c * EPC + e]);
                    if (mok) store16<T>(op + c * EPC, ov);
                    if (EPIF(EPI_ST, p.stats_out) || EPIF(EPI_CS, p.colstats)) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            v[c * EPC + e] = ov.get(e);
                            rs += ov.get(e);
                        }
                    }
                }
                if (EPIF(EPI_CS, p.colstats)) {
                    // GroupNorm statistics for the consumer of this tensor: (sum, sum of squares) per column over each 32-row block = the two
                    // 16-row MMA blocks 2h, 2h + 1 of this wave (i even: remember the row, i odd: add, reduce over the 16 lanes, store)
                    if ((i & 1) == 0) {
#pragma unroll
                        for (int e = 0; e < RUN; ++e) {
                            cs_a[e] = mok ? v[e] : 0.f;
                            cs_b[e] = mok ? v[e] * v[e] : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < RUN; ++e) {
                            cs_a[e] += mok ? v[e] : 0.f;
                            cs_b[e] += mok ? v[e] * v[e] : 0.f;
                        }
                        if constexpr (RUN == 8 || RUN == 16) colsum16<RUN>(cs_a, cs_b, c16);  // (other runs: probing tiles only, never launched with column statistics)
                        const int blk = (m0 + wm * WME + 16 * (i - 1)) >> 5;  // (tiles start on multiples of 64 rows)
                        if ((RUN == 16 || c16 < 8) && (blk << 5) < p.M) {
                            f32x2 st = {cs_a[0], cs_b[0]};
                            *reinterpret_cast<f32x2*>(p.colstats + ((int64_t)blk * p.N + n + (c16 & (RUN - 1))) * 2) = st;
                        }
                    }
                }
                if (EPIF(EPI_ST, p.stats_out)) {
                    // (mean, M2) of this lane's RUN columns, Chan-merged over the lane groups that share a 32-column chunk:
                    // RUN = 16 -> groups (g, g^1); RUN = 8 -> all four groups.  Lower group first on both sides: identical bits.
                    float mean = rs * (1.0f / RUN), m2 = 0.f, cn = (float)RUN;
#pragma unroll
                    for (int e = 0; e < RUN; ++e) m2 += (v[e] - mean) * (v[e] - mean);
                    constexpr int GPC = 32 / RUN;  // lane groups per 32-column chunk
#pragma unroll
                    for (int o = 16; o < 16 * GPC; o <<= 1) {
                        const float nb = __shfl_xor(cn, o), mb2 = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
                        if (lane & o) {
                            float n2 = nb, me2 = mb2, q2 = qb;
                            stat_merge(n2, me2, q2, cn, mean, m2);
                            cn = n2, mean = me2, m2 = q2;
                        } else {
                            stat_merge(cn, mean, m2, nb, mb2, qb);
                        }
                    }
                    if (mok && (g % GPC) == 0) {
                        const int chunk = n / 32;
                        f32x2 st = {mean, m2};
                        *reinterpret_cast<f32x2*>(p.stats_out + ((int64_t)chunk * p.M + m) * 2) = st;
                    }
                }
            }
        } else if (!FAST && mok) {  // (FAST: the caller guarantees aligned operands and N % 16 == 0 -- a lane's 16 columns are all inside N or all outside)
            // guarded scalar path (N edge tiles, unaligned outputs); geglu / LayerNorm fusion are never routed here (host checks)
#pragma unroll
            for (int e = 0; e < RUN; ++e) {
                const int nn = n + e;
                if (nn < p.N) {
                    float val = v[e];
                    if (bias) val += to_f32(bias[nn]);
                    if (rowbias) val += to_f32(rowbias[(int64_t)(m / p.rows_per_group) * p.ld_rowbias + nn]);
                    if (p.gelu) val = p.gelu == 1 ? gelu_exact(val) : quick_gelu(val);
                    if (res) val += to_f32(res[(int64_t)m * p.ldres + nn]);
                    if (p.out_f32) reinterpret_cast<float*>(p.out)[(int64_t)m * p.ldo + nn] = val;
                    else out[(int64_t)m * p.ldo + nn] = from_f32<T>(val);
                }
            }
        }
    }
    };
    if constexpr (FAST) {
        // one instance per kind of launch the UNet issues (the others take the general one): a workgroup that owns its CU alone pays every branch and every
        // re-loaded flag of the general row loop in full -- 1.15 us per row block of a 256 x 256 tile, 9.5 us per tile (profiles/r05_l_probe_g8_stamps.log)
        const unsigned have = (p.ln_stats ? EPI_LN : 0u) | (rowbias ? EPI_RB : 0u) | (p.gelu ? EPI_GELU : 0u) | (p.geglu ? EPI_GEGLU : 0u) | (res ? EPI_RES : 0u) | (p.out_f32 ? EPI_F32 : 0u) |
                              (p.stats_out ? EPI_ST : 0u) | (p.colstats ? EPI_CS : 0u);
        switch (have) {
            case 0u: rows(std::integral_constant<unsigned, 0u>{}); break;
            case EPI_LN: rows(std::integral_constant<unsigned, EPI_LN>{}); break;
            case EPI_LN | EPI_GEGLU: rows(std::integral_constant<unsigned, EPI_LN | EPI_GEGLU>{}); break;
            case EPI_RES | EPI_ST: rows(std::integral_constant<unsigned, EPI_RES | EPI_ST>{}); break;
            case EPI_CS: rows(std::integral_constant<unsigned, EPI_CS>{}); break;
            case EPI_RB | EPI_CS: rows(std::integral_constant<unsigned, EPI_RB | EPI_CS>{}); break;
            case EPI_RES | EPI_CS: rows(std::integral_constant<unsigned, EPI_RES | EPI_CS>{}); break;
            default: rows(std::integral_constant<unsigned, EPI_ALL>{}); break;
        }
    } else {
        rows(std::integral_constant<unsigned, EPI_ALL>{});
    }
    }
}

// out[m][n] = dtype( sum_s partial[s][m][n] (fixed order) + bias[n] + rowbias[m / rpg][n] + res[m][n] ).  One workgroup = 32 rows x 64 columns:
// thread (ty, tx) = (t >> 4, t & 15) owns rows 2 ty, 2 ty + 1 of the block at columns 4 tx .. 4 tx + 3 -- every partial it needs is requested up front
// (2 rows x ksplit independent 16-byte loads), (M / 32) x (N / 64) workgroups keep the memory system as busy as the old grid-stride form -- and the
// per-column sums GemmP::colstats asks for are 2 local adds + one fixed-order sum of the sixteen row pairs through LDS.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmP p) {
    __shared__ float red[15][16][8];
    constexpr int MAXS = 4;  // splits whose partials are all in flight at once (more: a serial tail)
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int nbx = (p.N + 63) / 64;
    const int bm = blockIdx.x / nbx, bn = blockIdx.x - bm * nbx;
    const int n = bn * 64 + tx * 4;
    T* out = reinterpret_cast<T*>(p.out);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowbias = reinterpret_cast<const T*>(p.rowbias);
    const T* res = reinterpret_cast<const T*>(p.res);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full4 = n + 4 <= p.N;
    if (n < p.N) {
        f32x4 part[2][MAXS];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = min(bm * 32 + ty * 2 + r, p.M - 1);
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                part[r][s] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (s < p.ksplit) {
                    const float* pp = p.partial + ((int64_t)s * p.M + m) * p.N + n;
                    if (full4) part[r][s] = *reinterpret_cast<const f32x4*>(pp);
                    else
                        for (int q = 0; q < 4; ++q)
                            if (n + q < p.N) part[r][s][q] = pp[q];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = bm * 32 + ty * 2 + r;
            if (m >= p.M) break;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < p.ksplit) v[0] += part[r][s][0], v[1] += part[r][s][1], v[2] += part[r][s][2], v[3] += part[r][s][3];
            for (int s = MAXS; s < p.ksplit; ++s) {
                const float* pp = p.partial + ((int64_t)s * p.M + m) * p.N + n;
                for (int q = 0; q < 4; ++q)
                    if (n + q < p.N) v[q] += pp[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nn = n + q;
                if (nn >= p.N) break;
                float val = v[q];
                if (bias) val += to_f32(bias[nn]);
                if (rowbias) val += to_f32(rowbias[(int64_t)(m / p.rows_per_group) * p.ld_rowbias + nn]);
                if (p.gelu) val = p.gelu == 1 ? gelu_exact(val) : quick_gelu(val);
                if (res) val += to_f32(res[(int64_t)m * p.ldres + nn]);
                const T o = from_f32<T>(val);
                out[(int64_t)m * p.ldo + nn] = o;
                const float f = to_f32(o);
                s1[q] += f;
                s2[q] += f * f;
            }
        }
    }
    if (p.colstats) {  // (workgroup-uniform)
        if (ty > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                red[ty - 1][tx][q] = s1[q];
                red[ty - 1][tx][4 + q] = s2[q];
            }
        }
        __syncthreads();
        if (ty == 0 && n < p.N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (n + q >= p.N) break;
                float a = s1[q], b = s2[q];
#pragma unroll
                for (int j = 0; j < 15; ++j) {
                    a += red[j][tx][q];
                    b += red[j][tx][4 + q];
                }
                f32x2 st = {a, b};
                *reinterpret_cast<f32x2*>(p.colstats + ((int64_t)bm * p.N + n + q) * 2) = st;
            }
        }
    }
}

extern int g_pf_blocks;  // default number of prefetch workgroups when the caller gives spans but no count (0 = prefetch off)
extern int g_pf_mode;    // 1 = plain loads, 2 = non-temporal
extern int g_tile;       // 0 = heuristic / caller's hint, 1..6 = force a tile configuration (probing / A-B runs)
extern int g_stages;     // 0 = heuristic / caller's hint, 2..4 = force the LDS pipeline depth
extern int g_lora_dbg;   // probing: see GemmP::lora_dbg


// Host side, common to every tile configuration: tile counts, how the 8 XCDs share the tile grid, prefetch workgroups.
inline void plan_grid(GemmP& q, int BM, int BN, bool conv, int kg) {
    const GemmP& p = q;
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
        kx += conv ? (double)sg.nkb / (sg.ksize * sg.ksize) : (double)sg.nkb;
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
    if (kg > 1) q.pf_blocks = (q.pf_blocks / 2 + 7) / 8 * 8;  // twice the threads per prefetch workgroup
    q.pf_mode = g_pf_mode;
}

}  // namespace mi355x
