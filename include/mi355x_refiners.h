/*
 * mi355x_refiners.h -- C ABI of libmi355x_refiners.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * refiners SDXL-UNet hot path.
 *
 * refiners (finegrain-ai/refiners @ 2025-06-14) has no FFI of its own: every FLOP of the hot path is a stock ATen
 * call made from a `fl.Chain` leaf.  Each entry point below therefore replaces one (or a fused group of) those call
 * sites; the citation on every function is the reference file:line whose arithmetic it takes over.  The host side
 * (refiners_amd/, Python, mirrors fluxion.Chain / Adapter.inject) is the only caller; INTEGRATION.md shows the
 * ctypes binding a refiners maintainer would add.
 *
 * Conventions
 *   - plain C: raw device pointers, int32/int64 sizes, strides in ELEMENTS, no torch / hip types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream).
 *   - all tensors are owned by the caller; outputs are pre-allocated; nothing is allocated, freed, synchronised or
 *     retained inside a call; kernels are enqueued on `stream` and the call returns immediately.
 *   - dtype selects the storage type of activations / weights / outputs (accumulation is always f32 on the matrix
 *     cores: v_mfma_f32_16x16x32_bf16 for MI355X_BF16, v_mfma_f32_16x16x4_f32 for MI355X_F32, the parity mode).
 *   - every 16-byte-vectorised pointer (all activations, weights, outputs) must be 16-byte aligned and its leading
 *     stride a multiple of 16 bytes, unless a field says otherwise.
 *   - return value: 0 on success, a negative MI355X_E* code otherwise (never throws, never aborts).
 *   - results are bit-reproducible run to run (no atomics-ordered reductions), as the reference's
 *     tests/foundationals/latent_diffusion/test_sd15_unet.py:21-37 requires.
 */
#ifndef MI355X_REFINERS_H
#define MI355X_REFINERS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_ABI_VERSION 7

enum { MI355X_F32 = 0, MI355X_BF16 = 1 };

enum {
    MI355X_OK = 0,
    MI355X_EDTYPE = -1,   /* unknown dtype */
    MI355X_ESHAPE = -2,   /* unsupported shape / stride / alignment for this kernel */
    MI355X_ELAUNCH = -3,  /* the HIP runtime refused the launch (hipGetLastError != hipSuccess) */
    MI355X_EARG = -4      /* NULL where a pointer is required, bad enum, ... */
};

/* Library / device probes (no reference counterpart). */
int mi355x_abi_version(void);
/* Writes a NUL-terminated description ("gfx950 ... CUs") of the current device; returns MI355X_OK or a negative code. */
int mi355x_device_info(char* buf, int32_t buflen);

/* ------------------------------------------------------------------------------------------------------------
 * mi355x_gemm -- out[M,N] = epilogue( sum_s X_s[M,K_s] . W_s[N,K_s]^T )
 *
 * Replaces, in one launch:
 *   fl.Linear                      src/refiners/fluxion/layers/linear.py:9-56      (F.linear, bias)
 *   fl.Conv2d 3x3 / 1x1            src/refiners/fluxion/layers/conv.py:6-61        (implicit GEMM over NHWC)
 *   LoraAdapter = Sum(target, loras) src/refiners/fluxion/adapters/lora.py:383-397 (second K segment = [x.A^T | s.B])
 *   fl.GLU(fl.GeLU())              src/refiners/fluxion/layers/activations.py:83-160 (geglu epilogue)
 *   fl.Residual / fl.Sum adds      src/refiners/fluxion/layers/chain.py:891-927    (res / rowbias epilogue)
 *   ResidualConcatenator cat       src/refiners/foundationals/latent_diffusion/unet.py:69-79 (two X segments)
 *   Upsample's nearest interpolate src/refiners/fluxion/layers/sampling.py:112-161 (ups = 2 gather)
 *
 * Up to three K segments are accumulated into the same output tile.  A segment is either
 *   conv == 0 : X_s rows are plain rows, row m at x + m*ldx, K_s = k elements (multiple of 128 bytes);
 *   conv == 1 : X_s is an NHWC image [B][H][W][k channels] (pixel stride ldx); output row m = (b, oy, ox) of a
 *               ksize x ksize cross-correlation with zero padding ksize/2, stride `stride`, applied to the image
 *               nearest-upsampled by `ups`; K_s = ksize*ksize*k ordered (ky, kx, channel); W_s rows are [N][K_s].
 * Weight rows W_s[n] must be "N-packed" by the caller exactly as refiners_amd.native.pack does (identity order
 * unless geglu, where value/gate rows are interleaved in groups of 32 so that one lane holds both).
 * Long-K / small-MN problems (the 32x32-resolution convolutions: 320 tiles, K = 11520) can be split along K (`ksplit`).
 * Epilogue order: + bias[n] ; + rowbias[(m / rows_per_group)*ld_rowbias + n] ; geglu: v = a * gelu_erf(g) ;
 * (or v = gelu_erf(v) when geglu == 2, v = v * sigmoid(1.702 v) when geglu == 3: fl.GeLU with approximation NONE / SIGMOID,
 * src/refiners/fluxion/layers/activations.py:83-125) ;
 * + res[m*ldres + n] ; convert to dtype ; store out[m*ldo + n].
 */
#define MI355X_MAX_SEG 3
#define MI355X_MAX_PREFETCH 2

typedef struct {
    const void* x;
    int64_t ldx;     /* row stride (conv == 0) or pixel stride (conv == 1), elements */
    const void* w;
    int64_t ldw;     /* weight row stride, elements */
    int32_t k;       /* conv == 0: K of the segment; conv == 1: channels of the segment */
    int32_t ksize;   /* conv == 1: 1 or 3 */
    int32_t stride;  /* conv == 1: 1 or 2 */
    int32_t ups;     /* conv == 1: 1 or 2 */
    int32_t H, W;    /* conv == 1: stored input height / width (before `ups`) */
    int32_t asym;    /* conv == 1: 0 = zero padding ksize/2 on every side; 1 = padding only after the last row / column, i.e.
                        F.pad(x, (0, 1, 0, 1)) followed by an unpadded conv (fl.Downsample(padding=0), layers/sampling.py:41-109) */
    int32_t kblocked; /* bit 0: w is stored K-BLOCKED, [K*sizeof/128][N rows][128 bytes] (ldw ignored): the N x 128-byte slab a K step
                         needs is contiguous.  global->LDS streaming slows down with the operand's row stride (10 TB/s at 2.5 KB rows,
                         5 TB/s at 10 KB, 4.5 TB/s at the 23-46 KB rows of a 3x3 conv's weights: profiles/r01_aa_probe_glds.log), and
                         weights are static, so the host re-lays them once.  bit 1: the same for x ([K blocks][M rows][128 B], conv == 0). */
} mi355x_gemm_seg;

typedef struct {
    int32_t dtype;
    int32_t M, N;      /* N = number of weight rows (for geglu: 2x the output width) */
    int32_t nseg;
    int32_t conv;      /* 0 or 1, applies to every segment */
    int32_t B, OH, OW; /* conv == 1: M == B*OH*OW */
    mi355x_gemm_seg seg[MI355X_MAX_SEG];
    void* out;
    int64_t ldo;
    const void* bias;       /* [N] or NULL */
    const void* rowbias;    /* [M / rows_per_group][ld_rowbias] or NULL (RangeAdapter2d time-embedding bias) */
    int64_t ld_rowbias;
    int32_t rows_per_group;
    int32_t geglu;          /* epilogue activation: 0 none, 1 GEGLU (value * gelu_erf(gate), packed rows), 2 gelu_erf, 3 quick-GELU on every column */
    const void* res;        /* [M][ldres] or NULL */
    int64_t ldres;
    const void* zeros;      /* >= 256 zero bytes in device memory; required when conv == 1 */
    int32_t tile;           /* 0 = let the library choose; 1: 128x128  2: 128x64  3: 64x128  4: 64x64 (M x N);
                               6: 128x128 computed by 8 waves in two K groups (even / odd K blocks, summed through LDS in a fixed order):
                               for launches with fewer output tiles than CUs;
                               7: 256x256 on the 8-wave / eight-phase main loop (one workgroup per CU, counted LDS-DMA waits, the two halves of the
                               workgroup one barrier apart): the large-grid shapes;  8: the same loop as a persistent "stream-K" launch -- one
                               workgroup per CU, each taking an equal share of (output tiles x K tiles), partial tiles summed in a fixed order
                               through sk_ws -- for launches whose 256x256 tiles do not fill a whole number of rounds (needs sk_ws / sk_flags;
                               falls back to 7 without them);  9: the same loop on 192x256 tiles (wave tile 96x64), whole tiles only and no transposed
                               part: for launches whose 256-row tiles leave CUs idle in their only round (M = 8192, N = 1280: 160 tiles / 215);
                               10: the same loop on 128x256 tiles (wave tile 64x64; bf16 GEMMs only, float32 / conv requests run on the library's choice):
                               780 TFLOP/s in a full round;  11: 192-row tiles for a whole number of rounds + 128-row tiles for a whole number of
                               rounds in ONE launch, for shapes that admit such a split on this GPU (a CFG pair's FF1, 2048 x 10240: 256 + 256 tiles on
                               256 CUs; bf16 GEMMs; anything else asking for it runs as 9): -7 % hot, not in the measured table.
                               In-launch LoRA on this loop: ONE column group of a plain one-segment GEMM without out_t,
                               as whole tiles (8 with LoRA runs as 7) -- t = x A^T comes from producer workgroups at the head of the grid (one per
                               32 rows, two to a workgroup for stacked ranks 32 / 64; the 4-wave tiles' producers with this launch's larger LDS ring),
                               every tile adds (t)(s B)^T after its K loop.  7 / 8 / 9 do not combine with other in-launch LoRA forms (several groups,
                               out_t, conv), a column group transposed from a column that is not a multiple of 256, or operands of 2 GB and more: such
                               launches run on the library's own choice among 1..4.  With ksplit > 1 (a caller that split K for want of tiles) the
                               split is dropped where the 8-wave loop takes the launch and kept, on the 128x128 tile, where it cannot */
    int32_t ksplit;         /* <= 1: no split; s > 1: s workgroups share each output tile's K range and write float32
                               partial sums into `ws`, a second launch adds them in a fixed order (deterministic) and
                               applies the epilogue.  Not combinable with geglu. */
    void* ws;               /* split-K scratch, >= ksplit * M * N * 4 bytes, 16-byte aligned (ignored unless ksplit > 1) */
    int64_t ws_bytes;
    /* Optional weight prefetch for LATER launches: up to MI355X_MAX_PREFETCH read-only byte spans (weights that launches after this one will stream).
       `prefetch_blocks` extra workgroups of this launch (rounded up to a multiple of 8; 0 = library default) touch one word per 64
       bytes so that those lines sit in the 256 MB Infinity Cache by the time they are needed: SDXL reads 5.1 GB of weights once
       per step, i.e. every kernel would otherwise start on HBM misses.  No effect on the result. */
    const void* prefetch[MI355X_MAX_PREFETCH];
    int64_t prefetch_bytes[MI355X_MAX_PREFETCH];
    int32_t prefetch_blocks;
    int32_t out_kblocked;   /* geglu == 1 only: store the [M][N/2] result K-blocked, [(N/2)*sizeof/128][M][128 bytes] (ldo ignored), ready to be
                               the x operand (kblocked bit 1) of the next GEMM -- FeedForward's second Linear reads 10 KB rows otherwise */
    int32_t stages;         /* LDS pipeline depth: 0 = let the library choose, 2..4 (tile 6: 2 only) */
    /* Transposed column group (conv == 0): when out_t != NULL, output columns n >= nt_begin (a multiple of 128) are written as
       out_t[(n - nt_begin) * ldt + m] (the V^T [C][B*L] layout mi355x_attention wants) and only columns < nt_begin go to `out`.
       One launch over the stacked weights [Wq; Wk; Wv] then yields Q | K row-major and V transposed: the three projections of
       Distribute(Linear, Linear, Linear) in src/refiners/fluxion/layers/attentions.py:205-316.  bias is applied to both groups;
       rowbias / res / geglu / ksplit are not available with out_t.  `out` may be NULL when nt_begin == 0. */
    int32_t nt_begin;
    void* out_t;
    int64_t ldt;
    /* LayerNorm folded into the GEMM that consumes its output (fl.LayerNorm, src/refiners/fluxion/layers/norm.py:13-60, followed by
       fl.Linear: the three Residual bodies of CrossAttentionBlock, latent_diffusion/cross_attention.py:25-73).  x is the tensor the
       LayerNorm would have read; the caller passes weights already scaled by gamma (W' = W . diag(gamma)) and
         ln_s[n] = sum_k W'[n][k],   ln_c[n] = sum_k beta[k] W[n][k] + bias[n]          (float32, packed like the weight rows)
       and the kernel computes out = rstd[m] * (x W'^T - mean[m] * ln_s) + ln_c, mean / rstd of row m over the LayerNorm width taken
       from `ln_stats`: ln_parts x M pairs (mean, M2 = sum of squared deviations) of consecutive 32-column chunks, float32, laid out
       [part][m][2] -- exactly what `stats_out` of the launch that PRODUCED x wrote.  bias must be NULL (it is inside ln_c). */
    const void* ln_stats;
    int32_t ln_parts;
    float ln_eps;
    const void* ln_s;
    const void* ln_c;
    /* Producer side: besides `out`, write per-row (mean, M2) of every 32-column chunk of the STORED (rounded) output row into
       stats_out[(n / 32) * M + m] (float32 pairs; N a multiple of 64, vectorisable epilogue, no geglu / ksplit / out_t). */
    void* stats_out;
    int32_t out_f32;        /* 1 = `out` is float32 whatever `dtype` says (ldo in float elements; 16-byte aligned rows): raw scores for
                               mi355x_softmax_rows.  Not combinable with geglu / stats_out / out_t / ksplit. */
    /* LoRA inside the parent launch -- LoraAdapter = Sum(target, *loras), src/refiners/fluxion/adapters/lora.py:383-397, with
       Lora = Chain(down, up, Multiply(scale)) (:14-60; LinearLora :269-322, Conv2dLora :325-380) -- for a stacked rank lora_r of 32, 64
       or 128 (zero-padded):
         out += T( x . A_g^T ) . lora_b[n]^T      g = column group of n (lora_nb[g] <= n, groups start on multiples of 128)
       lora_a[g]: the lora_r stacked down-projection rows of group g, K-BLOCKED: [K*sizeof/128][lora_r][128 bytes] (conv == 1: the down
       convolutions' weights packed like w, same kernel size / stride / padding as segment 0; the up convolutions must be 1x1);
       lora_b: [N][lora_r] row-major, the up-projections already multiplied by their scales (rows follow the same N-packing as w).
       The LoRAs adapt segment 0.  x A^T is computed ONCE per 32 rows, by extra workgroups at the head of the launch's grid (no column
       tile recomputes it; every group reads segment 0's x; which kind of workgroup -- a small producer per 32 rows and group, or one
       ordinary tile per row tile for all groups -- is the library's choice per launch), rounded to `dtype` (the reference's intermediate tensor),
       handed to the output tiles through
         lora_t     scratch, 128-BYTE aligned, >= groups * GS bytes with GS = M * lora_r * sizeof(dtype) rounded up to a multiple of 128 (group g's
                    rows start at byte g * GS: a cache line never holds rows of two groups or of two 32-row blocks),
         lora_flags int32[groups * ceil(M / 32) + 1], zeroed once by the caller and private to this call site (they keep the last epoch); the LAST
                    word is an error flag: the kernel sets it to 1 when a tile waited 2 s for a hand-over that never came (the output is then
                    undefined; nothing traps, the context survives),
         lora_epoch device pointer to an int32 whose value differs from every value left in lora_flags: increment it (mi355x_epoch_bump)
                    before each launch, or once per replay of a recorded program whose LoRA launches each own their flags,
       and multiplied against lora_b in the tiles' epilogue.  lora_b == NULL: off.  Not combinable with out_f32 or tile 6;
       with ksplit the first split carries the LoRA term.
       With ln_stats (x un-normalised): lora_a carries gamma like w does (A' = A . diag(gamma)) and the caller adds
         lora_ls[g][r] = sum_k A'_g[r][k],   lora_lc[g][r] = sum_k beta[k] A_g[r][k]        (float32, [groups][lora_r]). */
    const void* lora_a[3];
    int32_t lora_nb[3];
    int32_t lora_groups;
    int32_t lora_r;
    const void* lora_b;
    const void* lora_ls;
    const void* lora_lc;
    void* lora_t;
    int32_t* lora_flags;
    const int32_t* lora_epoch;
    /* GroupNorm statistics of the output, for the mi355x_groupnorm call that consumes it (ResidualBlock = Conv2d -> GroupNorm -> SiLU -> Conv2d,
       src/refiners/foundationals/latent_diffusion/unet.py:6-51: the statistics pass over the convolution's output disappears):
       colstats_out[(m / 32) * N + n] = (sum, sum of squares) of out[32 (m / 32) .. + 31][n] AS STORED (rounded to `dtype`), float32 pairs,
       >= ceil(M / 32) * N * 2 floats, 8-byte aligned.  Written by the epilogue (by the reduction pass when ksplit > 1), fixed summation order.
       Needs the vectorisable epilogue, N a multiple of 16; not combinable with geglu / out_t / out_f32; a request for tile 6 (two K groups) runs on
       tile 1 instead.  NULL: off. */
    float* colstats_out;
    /* tile 8 only.  sk_ws: sk_slots x 256 KB of float32 scratch (16-byte aligned) -- one slot per workgroup, so sk_slots bounds the number of
       workgroups (256 = one per CU of an MI355X); sk_flags: int32[sk_slots + 1], zeroed once by the caller: [0, sk_slots) are hand-off flags that
       the kernel leaves zero again, the last word is an error flag the kernel raises if a deposit never arrived (after 2 s).  Both may be shared by
       every launch that runs on one stream; launches that may run concurrently need their own. */
    void* sk_ws;
    int32_t* sk_flags;
    int32_t sk_slots;
} mi355x_gemm_args;

int mi355x_gemm(const mi355x_gemm_args* args, void* stream);
/* *epoch += 1 (skipping 0) on `stream`: the hand-off generation of mi355x_gemm's in-launch LoRA (see lora_epoch). */
int mi355x_epoch_bump(int32_t* epoch, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * mi355x_attention -- out = sum_s out_scale_s * softmax(scale * Q K_s^T) V_s      (flash style, no mask, non causal)
 *
 * Replaces ScaledDotProductAttention.forward (src/refiners/fluxion/layers/attentions.py:60-202: head split :177-192,
 * F.scaled_dot_product_attention :15-34, head merge :194-202) and, with nstream == 2, the IP-Adapter
 * Sum(SDPA, ImageCrossAttention) of src/refiners/foundationals/latent_diffusion/image_prompt.py:237-309.
 *
 * Q, K, out are token-major: element (b, token, h*D + d) at base + b*batch_stride + token*ld + h*D + d.
 * V is passed TRANSPOSED (produced that way by mi355x_gemm with the operands swapped): element (h*D + d, b, key) at
 * vt + (h*D + d)*ldvt + b*vt_batch_stride + key; every V^T row must be readable (and finite) up to the next
 * multiple of 64 keys.  D must be 64 (every SDXL attention); other head shapes go through mi355x_attention_general.
 * Kernels behind it (csrc/attention.hip): at most three 64-key tiles over all streams (the step's cross-attentions) -- attn_short_dma_kernel (bf16) /
 * attn_short_kernel (float32); more tiles, bf16, one stream (the self-attentions) -- attn_pipe_kernel, whose running maximum is LAZY (a tile whose scores stay
 * within 2^8 of the reference is exponentiated against the old reference: same quotient O / l, last-digit differences against a per-tile maximum) and whose
 * row sums are taken over the bf16-rounded P; float32 (the 1e-3 parity mode) and two-stream launches with more tiles -- attn_kernel with a per-tile maximum.
 */
typedef struct {
    const void* k;
    int64_t ldk;
    int64_t k_batch_stride;
    const void* vt;
    int64_t ldvt;
    int64_t vt_batch_stride;
    int32_t Lk;
    float out_scale;
} mi355x_kv_stream;

typedef struct {
    int32_t dtype;
    int32_t B, H, D, Lq;
    int32_t nstream; /* 1 or 2 */
    const void* q;
    int64_t ldq;
    int64_t q_batch_stride;
    void* out;
    int64_t ldo;
    int64_t o_batch_stride;
    float scale;
    mi355x_kv_stream kv[2];
} mi355x_attn_args;

int mi355x_attention(const mi355x_attn_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * mi355x_attention_general -- out = out_scale * softmax(scale * Q K^T [+ causal mask]) V for head shapes other than 64:
 * a QK width Dqk and a V width Dv that need not be equal, one K/V stream.
 *
 * Replaces the same ScaledDotProductAttention.forward (src/refiners/fluxion/layers/attentions.py:60-202) where the head
 * dimension is not 64: SD1.5's 8 heads over 320/640/1280 channels (Dqk = Dv = 40/80/160,
 * src/refiners/foundationals/latent_diffusion/stable_diffusion_1/unet.py:30-45), `is_causal=True`
 * (attentions.py:15-34; key j attends to query i iff j <= i, both counted inside the sample), and the SegmentAnything
 * ViT attention with decomposed relative position bias (src/refiners/foundationals/segment_anything/image_encoder.py:
 * 82-127): the host appends the per-query bias rows (rel_h | rel_w) to Q / scale and one-hot row/column indicators to K,
 * so that Q'K'^T = QK^T + bias / scale exactly; then Dqk = 80 + h + w (padded to a multiple of 8 elements) and Dv = 80.
 *
 * Layouts as for mi355x_attention (Q, K, out token-major; V transposed, rows readable and finite up to the next
 * multiple of 64 keys).  Dqk * sizeof(dtype) % 16 == 0, Dv % 4 == 0; supported (Dqk, Dv) up to (64,64), (96,80),
 * (128,80), (160,160), (224,80); anything else returns MI355X_ESHAPE.
 */
typedef struct {
    int32_t dtype;
    int32_t B, H, Lq, Lk;
    int32_t Dqk, Dv;
    int32_t causal;
    const void* q;
    int64_t ldq;
    int64_t q_batch_stride;
    const void* k;
    int64_t ldk;
    int64_t k_batch_stride;
    const void* vt;
    int64_t ldvt;
    int64_t vt_batch_stride;
    void* out;
    int64_t ldo;
    int64_t o_batch_stride;
    float scale;
    float out_scale;
} mi355x_attn_general_args;

int mi355x_attention_general(const mi355x_attn_general_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * mi355x_layernorm -- rows of C: out = (x - mean) * rsqrt(var + eps) * gamma + beta
 * Replaces fl.LayerNorm (src/refiners/fluxion/layers/norm.py:13-46 -> F.layer_norm). C*sizeof(dtype) % 16 == 0,
 * C <= 4096.
 */
typedef struct {
    int32_t dtype;
    int32_t M, C;
    const void* x;
    int64_t ldx;
    const void* gamma;
    const void* beta;
    float eps;
    void* out;
    int64_t ldo;
} mi355x_layernorm_args;

int mi355x_layernorm(const mi355x_layernorm_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * mi355x_groupnorm -- NHWC GroupNorm (+ optional SiLU), three launches (partial sums, finalize, apply); TWO when the launch that
 * produced x also wrote its column statistics (`colstats` = that launch's mi355x_gemm_args.colstats_out; HW must be a multiple of 32).
 * Replaces fl.GroupNorm (src/refiners/fluxion/layers/norm.py:49-93 -> F.group_norm) and the fl.SiLU that follows it
 * in ResidualBlock (src/refiners/foundationals/latent_diffusion/unet.py:29-44) / OutputBlock.
 * x, out: [B][HW][C] with pixel stride ldx / ldo.  `ws` is float scratch of at least mi355x_groupnorm_ws_floats()
 * elements.  Statistics use a per-channel pivot and a fixed reduction order (deterministic, no atomics).
 */
typedef struct {
    int32_t dtype;
    int32_t B, HW, C, G;
    const void* x;
    int64_t ldx;
    const void* gamma;
    const void* beta;
    float eps;
    int32_t silu;
    void* out;
    int64_t ldo;
    float* ws;
    const float* colstats; /* [B * HW / 32][C][2] (sum, sum of squares) per 32-pixel block, or NULL (statistics are then computed from x) */
    /* Two-source form: the normalised tensor is Concatenate(x, x2) along the channels WITHOUT existing (ResidualConcatenator -> ResidualBlock,
       src/refiners/foundationals/latent_diffusion/unet.py:69-85): channels [0, C1) are read from x, [C1, C) from x2 (pixel stride ldx2).  C1 * sizeof(dtype)
       a multiple of 16.  With colstats: colstats is [B * HW / 32][C1][2] and colstats2 [B * HW / 32][C - C1][2] (each source's producer wrote its own).
       x2 == NULL: one source. */
    const void* x2;
    int64_t ldx2;
    int32_t C1;
    const float* colstats2;
} mi355x_groupnorm_args;

int64_t mi355x_groupnorm_ws_floats(int32_t B, int32_t HW, int32_t C);
int mi355x_groupnorm(const mi355x_groupnorm_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Layout / glue kernels (HBM-bound, 16-byte vectorised).
 */
/* NCHW [B][C][HW] -> NHWC [B][HW][ldo] (first `C` channels of each pixel are written). */
int mi355x_nchw_to_nhwc(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t HW, int64_t ldo, void* stream);
/* NHWC [B][HW][ldx] -> NCHW [B][C][HW]. */
int mi355x_nhwc_to_nchw(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t HW, int64_t ldx, void* stream);
/* im2col of a small-channel NCHW image for the UNet's first 3x3 conv (src/refiners/foundationals/latent_diffusion/
 * stable_diffusion_xl/unet.py:118-121): out[M = B*H*W][ldo], column (ky*3+kx)*C + c, zero elsewhere up to ldo. */
int mi355x_im2col3x3_nchw(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int64_t ldo, void* stream);
/* Non-overlapping P x P patches of an NCHW image as GEMM rows (SAM's PatchEncoder convolution, src/refiners/foundationals/
 * segment_anything/image_encoder.py:8-44): out[(b, py, px)][c*P*P + ky*P + kx] = x[b][c][py*P + ky][px*P + kx]. */
int mi355x_patchify_nchw(int32_t dtype, const void* x, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t P, int64_t ldo, void* stream);
/* out[i][0:C] = idx[i] >= 0 ? x[idx[i]][0:C] : 0   (row gather with zero fill: WindowPartition / WindowMerge of
 * segment_anything/image_encoder.py:202-236 as static index tables).  C * sizeof(dtype) must be a multiple of 16. */
int mi355x_gather_rows(int32_t dtype, const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int64_t n_rows, int32_t C, void* stream);
/* SegmentAnything's decomposed relative position bias (src/refiners/foundationals/segment_anything/image_encoder.py:82-127)
 * as extra query columns.  src row per head: [q*scale (d) | P1 (2*S1-1) | P2 (2*S2-1) | pad] (Lp columns), P1[r] = q.E1[2*S1-2-r],
 * P2[r] = q.E2[2*S2-2-r] (tables folded into the projection weights by the host).  Token t of a sample is at (a, b) = (t / S2, t % S2).
 * out row per head: [q*scale (d) | P1[S1-1-a : 2*S1-1-a] | P2[S2-1-b : 2*S2-1-b] | 0] (Dq columns): against K' = [k | onehot(a') |
 * onehot(b') | 0] the product is scale*q.k + q.E1[a-a'+S1-1] + q.E2[b-b'+S2-1], the logits of RelativePositionAttention. */
int mi355x_relpos_pack(int32_t dtype, const void* src, int64_t lds, void* out, int64_t ldo, int64_t M, int32_t H, int32_t d, int32_t S1, int32_t S2,
                       int32_t Lp, int32_t Dq, void* stream);
/* 1x1 convolution of an NCHW image with at most 8 input and 8 output channels (the VAE decoder's 4 -> 4 conv on the latents,
 * src/refiners/foundationals/latent_diffusion/auto_encoder.py:185-187): out[b][o][p] = bias[o] + sum_c w[o][c] x[b][c][p]. */
int mi355x_pointwise_nchw(int32_t dtype, const void* x, const void* w, const void* bias, void* out, int32_t B, int32_t Ci, int32_t Co,
                          int64_t HW, void* stream);
/* out[m][0:C1] = a[m][0:C1]; out[m][C1:C1+C2] = b[m][0:C2]   (ResidualConcatenator, unet.py:69-79, NHWC). */
int mi355x_concat2(int32_t dtype, const void* a, int64_t lda, int32_t C1, const void* b, int64_t ldb, int32_t C2,
                   void* out, int64_t ldo, int64_t M, void* stream);
/* out = alpha*a + beta*b elementwise over n elements (fl.Sum / Residual glue, ControlLora residual injection). */
int mi355x_axpby(int32_t dtype, const void* a, float alpha, const void* b, float beta, void* out, int64_t n, void* stream);
/* Row softmax for attention heads too wide for the flash kernels (the SDXL VAE's single 512-wide head over H*W tokens,
 * src/refiners/foundationals/latent_diffusion/auto_encoder.py:108,175 -> fluxion/layers/attentions.py:388): with
 *   S = Q K^T (mi355x_gemm, out_f32)  ->  P = mi355x_softmax_rows(S)  ->  O = P V (mi355x_gemm with V^T as the weight operand)
 * out[m][j] = exp(scale * (s[m][j] - max_j s[m][j])) / sum_j ... for j < L, and exactly 0 for L <= j < Lp (the K padding of
 * the second GEMM).  s: float32 rows of stride lds; out: `dtype` rows of stride ldo.  Bit-reproducible. */
int mi355x_softmax_rows(int32_t dtype, const float* s, int64_t lds, void* out, int64_t ldo, int64_t M, int32_t L, int32_t Lp, float scale,
                        void* stream);
/* Self-Attention Guidance (src/refiners/foundationals/latent_diffusion/self_attention_guidance.py:22-105, xl/model.py:164-250).
 * mi355x_colsum_rows: acc[j] (+)= scale * sum_{i < M} p[i*ldp + j], j < L  -- the attention mass a key receives from all queries of
 *   one head (p = that head's softmax probabilities from mi355x_softmax_rows); heads are summed by consecutive launches with
 *   accumulate = 1 and scale = 1 / heads, giving attn_map.mean(dim=1).sum(dim=1) of SAGAdapter.compute_sag_mask (:72-84).
 * mi355x_sag_degrade: compute_degraded_latents (:86-95) in one kernel: x0 = (x - coef[2]*eps) / coef[1] (Solver.remove_noise), k x k
 *   Gaussian blur with reflect padding (separable weights w1[ksize], fluxion/utils.py:65-113) where mass[b][cell] > 1 (cell = the
 *   nearest-neighbour (ah, aw) attention cell of the pixel), then coef[1] * (.) + coef[2] * eps (Solver.add_noise).
 *   x, eps, out: [n][C][h][w] contiguous; mass: float32 [n][ah*aw]; coef: device floats, the step's row of the CFG+DDIM table. */
int mi355x_colsum_rows(int32_t dtype, const void* p, int64_t ldp, int32_t M, int32_t L, float* acc, int32_t accumulate, float scale, void* stream);
int mi355x_sag_degrade(int32_t dtype, const void* x, const void* eps, const float* mass, int32_t ah, int32_t aw, const float* coef, const float* w1,
                       int32_t ksize, void* out, int32_t n, int32_t C, int32_t h, int32_t w, void* stream);
/* out = silu(x) over n elements. */
int mi355x_silu(int32_t dtype, const void* x, void* out, int64_t n, void* stream);

/* Sinusoidal embedding (src/refiners/foundationals/latent_diffusion/range_adapter.py:11-22, used for the timestep,
 * stable_diffusion_xl/unet.py:56-78, and for time_ids, :20-53): for each of the n float32 values x[i],
 *   out[(i / group) * ldo + col0 + (i % group) * dim + j]           = cos(x[i] * 10000^(-j / (dim/2)))
 *   out[(i / group) * ldo + col0 + (i % group) * dim + dim/2 + j]   = sin(x[i] * 10000^(-j / (dim/2))),  j < dim/2
 * computed in float32 and stored as `dtype`.  group = values per output row (1 for the timestep, 6 for SDXL's time_ids). */
int mi355x_sinusoidal(int32_t dtype, const float* x, int64_t n, int32_t dim, int32_t group, void* out, int64_t ldo, int32_t col0,
                      void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * mi355x_cfg_ddim_step -- classifier-free-guidance combine + DDIM update in one launch, no host sync.
 * Replaces LatentDiffusionModel.forward's chunk/combine (src/refiners/foundationals/latent_diffusion/model.py:142-145)
 * and DDIM.__call__ (src/refiners/foundationals/latent_diffusion/solvers/ddim.py:56-95):
 *   eps = uncond + cfg*(cond - uncond);  x0 = (x - sqrt(1-a_t)*eps)/sqrt(a_t);  x' = sqrt(a_prev)*x0 + sqrt(1-a_prev)*eps
 * `unet_out` holds [uncond ; cond] as two consecutive blocks of n elements.  coef = {cfg, sqrt(a_t), sqrt(1-a_t),
 * sqrt(a_prev), sqrt(1-a_prev)} is read from DEVICE memory (f32[5]) so that a captured graph can be replayed with new
 * step coefficients.  x is updated in place (f32 or bf16 per dtype; arithmetic in f32).
 */
int mi355x_cfg_ddim_step(int32_t dtype, void* x, const void* unet_out, const float* coef, int64_t n, void* stream);
/* Classifier-free guidance + one step of any solver that is linear in (x, eps, one kept quantity) -- Euler
 * (src/refiners/foundationals/latent_diffusion/solvers/euler.py:62-100), DPM-Solver++ 2M with sde_variance 0 (solvers/dpm.py:224-329),
 * DDIM -- after the CFG combine of latent_diffusion/model.py:142-145:
 *   eps = u + cfg (c - u);  d = hx x + he eps;  x' = kx x + ke eps + kd d + kp hist;  hist = d;
 *   model_in[0:n] = model_in[n:2n] = s_next x'   (Solver.scale_model_input of the NEXT step on cat(x', x'); may be NULL)
 * coef = {cfg, hx, he, kx, ke, kd, kp, s_next}, eight floats in DEVICE memory; x, hist: n elements; unet_out: 2n (u then c). */
int mi355x_cfg_linear_step(int32_t dtype, void* x, const void* unet_out, void* hist, void* model_in, const float* coef, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_REFINERS_H */
