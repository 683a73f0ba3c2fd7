// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the refiners hot path.
//
// Conventions used by every kernel in this directory
//   * wavefront = 64 lanes; "g" = lane >> 4 (16-lane group), "c16" = lane & 15.
//   * operand fragments are 16-byte chunks ("frag_t"): 8 bf16 or 4 f32 consecutive along K.
//   * one "MMA step" multiplies two 16-row fragment sets over 4 chunks of K (one chunk per lane group):
//       bf16 : 1 x v_mfma_f32_16x16x32_bf16          (K = 32)
//       f32  : 4 x v_mfma_f32_16x16x4_f32            (K = 16, element e of the chunk feeds MFMA e)
//     so the SAME LDS image / fragment addressing serves both dtypes (only the bytes per element differ).
//   * MFMA operand roles: A rows come from the first fragment, B columns from the second; result lane layout
//       D[row = 4*g + r][col = c16],  r = 0..3  (MI355X guide, cdna_hip_programming.md section 3).
//   * LDS tiles are rows of 128 B or 256 B, 16-B chunks XOR-swizzled by row so that ds_read_b128 fragment reads and
//     ds_read_b64 half-chunk reads are bank-conflict free; the global->LDS copy is `global_load_lds_dwordx4`
//     (LDS destination lane-linear), so the swizzle is applied to the per-lane SOURCE address (guide rule 21).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_DEV __device__ __forceinline__

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) int frag_t;   // one 16-byte operand chunk

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int KSTEP = 16;  // K covered by one MMA step (4 chunks)
};
template <> struct DT<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int KSTEP = 32;
};

MI_DEV float to_f32(float v) { return v; }
MI_DEV float to_f32(bf16_t v) { return (float)v; }
template <typename T> MI_DEV T from_f32(float v);
template <> MI_DEV float from_f32<float>(float v) { return v; }
template <> MI_DEV bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// One MMA step: acc += A(16 rows x Kstep) * B(16 cols x Kstep)^T, fragments as described above.
template <typename T> MI_DEV void mma_step(f32x4& acc, frag_t a, frag_t b);
template <> MI_DEV void mma_step<bf16_t>(f32x4& acc, frag_t a, frag_t b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> MI_DEV void mma_step<float>(f32x4& acc, frag_t a, frag_t b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
}

// ---- swizzled LDS tiles ------------------------------------------------------------------------------------
// A tile is `rows` rows of ROWB bytes (ROWB = 128 or 256), i.e. CPR = ROWB/16 chunks per row.
// Physical chunk = logical chunk XOR swz(row).  For ROWB = 128 two rows share one 256-B bank row, so the swizzle uses
// row>>1 (3 bits); for ROWB = 256 it uses row (4 bits).  Either way 16 consecutive rows at one logical chunk hit 16
// distinct 16-B slots of the 64-bank window.
template <int ROWB> MI_DEV int swz(int row) {
    if constexpr (ROWB == 128) return (row >> 1) & 7;
    else return row & 15;
}
template <int ROWB> MI_DEV int tile_off(int row, int chunk) {  // byte offset of logical (row, chunk)
    return row * ROWB + ((chunk ^ swz<ROWB>(row)) << 4);
}

// Key held by LDS row `row` of a 64-key K tile (bf16): LDS row 16 t + r16 holds key 32 (t >> 1) + 8 (r16 >> 2) + 4 (t & 1) + (r16 & 3).
// S^T block t then gives lane group g the keys 8 g + 4 (t & 1) + {0..3} of its 32-key half, so the P^T fragment a lane assembles from
// blocks 2 s, 2 s + 1 covers EIGHT CONSECUTIVE keys 32 s + 8 g .. + 7 and the matching V^T fragment is ONE 16-byte chunk (4 s + g) of the
// row -- read with the GEMM's conflict-free ds_read_b128 pattern.  With keys in natural order the fragment was two 8-byte halves two chunks
// apart: ds_read_b64 pairs that ran at a 2-way bank conflict (a third of the kernel's LDS cycles, profiles/r03_q_pmc_sq_by_kernel.json).
// The permutation lives in the loader's SOURCE row (and in the tail / causal masks); LDS addressing of the K reads is unchanged.
// Used by attn_kernel, attn_short_kernel (attention.hip) and attn_general_kernel (attention_general.hip).
MI_DEV constexpr int k_row_key(int row) { return (row & 32) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3); }

MI_DEV frag_t lds_read_frag(const char* lds, int byte_off) {
    return *reinterpret_cast<const frag_t*>(lds + byte_off);
}

// Asynchronous 16-byte-per-lane global -> LDS copy.  `lds_wave_base` must be wave-uniform: lane i lands at
// lds_wave_base + 16*i.  Completion is tracked by vmcnt (wait with wait_vm0()).
MI_DEV void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_wave_base, 16, 0, 0);
}
MI_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> MI_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

MI_DEV int lane_id() { return threadIdx.x & 63; }
MI_DEV int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// Bijective XCD-aware remap of a 1-D block id (guide 5.5 T1): blocks that are adjacent after the remap run on the
// same XCD (hardware places block b on XCD b % 8) and therefore share that XCD's L2.
MI_DEV int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// v_exp_f32 without the denormal-range fix-up sequence exp2f() expands to: softmax arguments are <= 0 and results that
// would be denormal are flushed to 0, which is what an online softmax wants.
MI_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// erf-GELU (fl.GeLU with approximation NONE): erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. float32 round-off level:
// two transcendentals + a degree-5 Horner instead of libm's erff, which costs 3x the VALU work of the GEGLU epilogue's 32 calls per lane)
MI_DEV float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float y = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
    return copysignf(y, x);
}
MI_DEV float gelu_exact(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// CLIP's "quick GELU" (GeLUApproximation.SIGMOID, fluxion/layers/activations.py:83-118)
MI_DEV float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// x * sigmoid(x) with ONE v_rcp_f32 (1 ulp) instead of the IEEE division's ten instructions: the GroupNorm apply pass is 8 SiLUs per 16 bytes, and at the
// step's sizes (Infinity-Cache resident tensors) the division was a sixth of it: 18.8 -> 16.4 us per GroupNorm at 2 x 128 x 128 x 320 (profiles/r06_zh_probe_gn_fast_silu.log)
MI_DEV float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// 16-byte vector of T, for epilogues and elementwise kernels.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    f32x4 v;
    static constexpr int N = 4;
    MI_DEV float get(int i) const { return v[i]; }
    MI_DEV void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    bf16x8 v;
    static constexpr int N = 8;
    MI_DEV float get(int i) const { return (float)v[i]; }
    MI_DEV void set(int i, float x) { v[i] = (bf16_t)x; }
};
template <typename T> MI_DEV Vec16<T> load16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> MI_DEV void store16(T* p, const Vec16<T>& x) {
#ifdef MI355X_ABL_NOSTORE  // probing builds only: the value is computed, nothing is written
    asm volatile("" ::"v"(x.v), "v"(p));
#else
    *reinterpret_cast<decltype(x.v)*>(p) = x.v;
#endif
}
