// mi355x_gemm: C entry point, argument validation and the plain-GEMM instantiations of gemm_kernel.cuh.
// (The implicit-GEMM convolution instantiations live in gemm_conv.hip so that the two halves compile in parallel.)
#include <climits>

#include "gemm8_kernel.cuh"
#include "gemm_kernel.cuh"

namespace mi355x {

int g_pf_blocks = 64;
int g_pf_mode = 1;
int g_tile = 0;
int g_stages = 0;
int g_lora_dbg = 0;
int g_sk_g = 0;
int g_g8_persist = 1;

namespace {
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

}  // namespace mi355x

using namespace mi355x;

extern "C" int mi355x_set_option(const char* name, int value);
extern "C" int mi355x_set_option(const char* name, int value) {
    // debugging / A-B switches; not part of the stable contract
    if (!name) return MI355X_EARG;
    if (name[0] == 'g' && name[1] == 'n') {  // "gnwgs" / "gnunroll" (norm.hip)
        extern int g_gn_wgs, g_gn_unroll;
        if (name[2] == 'w') g_gn_wgs = value;
        else g_gn_unroll = value;
        return MI355X_OK;
    }
    if (name[0] == 'g' && name[1] == '8') {  // "g8persist"
        g_g8_persist = value;
        return MI355X_OK;
    }
    if (name[0] == 's' && name[1] == 'k') {  // "skg": number of stream-K workgroups (0 = one per CU)
        g_sk_g = value;
        return MI355X_OK;
    }
    if (name[0] == 's') {  // "stages"
        g_stages = value;
        return MI355X_OK;
    }
    if (name[0] == 'p') {  // "pfblocks" / "pfmode"
        if (name[2] == 'b') g_pf_blocks = value < 0 ? 0 : (value + 7) / 8 * 8;
        else g_pf_mode = value;
        return MI355X_OK;
    }
    if (name[0] == 'l') {  // "lora_dbg"
        g_lora_dbg = value;
        return MI355X_OK;
    }
    if (name[0] == 't') {  // "tile"
        g_tile = value;
        return MI355X_OK;
    }
    return MI355X_EARG;
}

static int g_stat_g8 = 0, g_stat_g8_lora = 0, g_stat_g9 = 0, g_stat_g11 = 0;
extern "C" int mi355x_get_stat(const char* name);
extern "C" int mi355x_get_stat(const char* name) {
    // launches since the library was loaded (tests: did the configuration asked for really run?); like mi355x_set_option not part of the stable contract
    //   "g8" = launches on the 8-wave loop (tile configurations 7 / 8 / 9), "g8lora" = those of them with the in-launch LoRA, "g9" = those on 192-row tiles
    if (!name) return MI355X_EARG;
    if (name[0] == 'g' && name[1] == '8') return name[2] == 'l' ? g_stat_g8_lora : g_stat_g8;
    if (name[0] == 'g' && name[1] == '9') return g_stat_g9;
    if (name[0] == 'g' && name[1] == '1' && name[2] == '1') return g_stat_g11;
    return MI355X_EARG;
}

namespace {
__global__ void epoch_bump_kernel(int* e) {
    const int v = *e + 1;
    *e = v == 0 ? 1 : v;  // 0 is what freshly zeroed flags hold: never a valid epoch
}
}  // namespace

extern "C" int mi355x_epoch_bump(int32_t* epoch, void* stream) {
    if (!epoch) return MI355X_EARG;
    hipLaunchKernelGGL(epoch_bump_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), epoch);
    return hipGetLastError() == hipSuccess ? MI355X_OK : MI355X_ELAUNCH;
}

extern "C" int mi355x_gemm(const mi355x_gemm_args* a, void* stream) {
    if (!a) return MI355X_EARG;
    const bool has_t = a->out_t != nullptr;
    if (!a->out && !(has_t && a->nt_begin == 0)) return MI355X_EARG;
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
        {  // operand extents from x / w (the 8-phase loop's buffer descriptors)
            const int64_t taps = a->conv ? (int64_t)g.ksize * g.ksize : 1, kbytes = (int64_t)g.k * es, nkb_all = taps * (g.k / bke);
            const int64_t xrows = a->conv ? (int64_t)a->B * g.H * g.W : (int64_t)a->M;
            d.xbytes = xkb ? nkb_all * a->M * 128 : (xrows - 1) * d.ldxb + kbytes;
            d.wbytes = wkb ? nkb_all * a->N * 128 : ((int64_t)a->N - 1) * d.ldwb + taps * kbytes;
        }
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
    const int eso = a->out_f32 ? 4 : es;
    bool vec = (!a->out || (aligned16(a->out) && (a->ldo * eso) % 16 == 0));
    if (a->bias) vec = vec && aligned16(a->bias);
    if (a->rowbias) vec = vec && aligned16(a->rowbias) && (a->ld_rowbias * es) % 16 == 0;
    if (a->res) vec = vec && aligned16(a->res) && (a->ldres * es) % 16 == 0;
    if (has_t) vec = vec && aligned16(a->out_t) && (a->ldt * es) % 16 == 0;
    p.vec_ok = vec ? 1 : 0;
    if (p.geglu && (!vec || a->N % 64)) return MI355X_ESHAPE;
    p.out_kb = a->out_kblocked ? 1 : 0;
    if (p.out_kb && (!p.geglu || a->res || a->N % 256 || a->ksplit > 1)) return MI355X_ESHAPE;  // only the fused-GEGLU store path writes it
    p.tile_hint = a->tile;
    p.stage_hint = a->stages;
    // transposed column group
    p.nt_begin = INT_MAX;
    if (has_t) {
        if (a->conv || a->ksplit > 1 || a->nt_begin < 0 || a->nt_begin > a->N || a->nt_begin % 128 || a->geglu || a->rowbias || a->res || a->ldt < a->M) return MI355X_ESHAPE;
        p.nt_begin = a->nt_begin;
        p.out_t = static_cast<char*>(a->out_t);
        p.ldt = a->ldt;
    }
    // LayerNorm fusion
    if (a->ln_stats) {
        if (a->conv || a->ksplit > 1 || !a->ln_s || !a->ln_c || a->ln_parts <= 0 || a->bias || !vec || a->N % 64 || !aligned16(a->ln_s) || !aligned16(a->ln_c) ||
            (reinterpret_cast<uintptr_t>(a->ln_stats) & 7))
            return MI355X_ESHAPE;
        p.ln_stats = static_cast<const float*>(a->ln_stats);
        p.ln_parts = a->ln_parts;
        p.ln_eps = a->ln_eps;
        p.ln_s = static_cast<const float*>(a->ln_s);
        p.ln_c = static_cast<const float*>(a->ln_c);
    }
    if (a->lora_b) {
        if (a->ksplit > 1 && a->geglu) return MI355X_ESHAPE;
        if ((a->ln_stats && (!a->lora_ls || !a->lora_lc)) || a->out_f32 || a->lora_groups < 1 || a->lora_groups > 3 || a->lora_nb[0] != 0 ||
            !aligned16(a->lora_b) || (a->lora_r != 32 && a->lora_r != 64 && a->lora_r != 128) || !a->lora_t || (reinterpret_cast<uintptr_t>(a->lora_t) & 127) || !a->lora_flags ||
            !a->lora_epoch || (a->conv && a->lora_groups != 1))
            return MI355X_ESHAPE;
        for (int g = 0; g < a->lora_groups; ++g) {
            if (!a->lora_a[g] || !aligned16(a->lora_a[g]) || a->lora_nb[g] % 128 || (g && a->lora_nb[g] <= a->lora_nb[g - 1])) return MI355X_ESHAPE;
            p.lora_a[g] = static_cast<const char*>(a->lora_a[g]);
            p.lora_nb[g] = a->lora_nb[g];
        }
        p.lora_groups = a->lora_groups;
        p.lora_r = a->lora_r;
        p.lora_b = static_cast<const char*>(a->lora_b);
        p.lora_ls = static_cast<const float*>(a->lora_ls);
        p.lora_lc = static_cast<const float*>(a->lora_lc);
        p.lora_t = static_cast<char*>(a->lora_t);
        // group stride of the hand-off scratch: whole 128-byte lines per group, so that no cache line holds rows of two (group, row block) units
        // (the tiles read t with plain cacheable loads: a line may only be fetched after ITS flag was seen -- round-3 advisor finding)
        p.lora_gs = ((int64_t)a->M * a->lora_r * (a->dtype == MI355X_F32 ? 4 : 2) + 127) / 128 * 128;
        p.lora_flags = a->lora_flags;
        p.lora_epoch = a->lora_epoch;
    }
    if (a->out_f32) {
        if (a->conv || a->ksplit > 1 || a->geglu == 1 || has_t || a->stats_out || !a->out) return MI355X_ESHAPE;
        p.out_f32 = 1;
    }
    if (a->stats_out) {
        if (a->conv || a->ksplit > 1 || a->geglu == 1 || has_t || !vec || a->N % 64 || (reinterpret_cast<uintptr_t>(a->stats_out) & 7)) return MI355X_ESHAPE;
        p.stats_out = static_cast<float*>(a->stats_out);
    }
    if (a->colstats_out) {
        if (a->geglu == 1 || has_t || a->out_f32 || !vec || a->N % 16 || (reinterpret_cast<uintptr_t>(a->colstats_out) & 7)) return MI355X_ESHAPE;  // (the two-K-group tile has no statistics epilogue: launch_tile takes 128 x 128 instead)
        p.colstats = a->colstats_out;
    }
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
    const int tile_req = g_tile ? g_tile : a->tile;
    int g8_mt = tile_req == 9 ? 6 : tile_req == 10 ? 4 : 8;  // tile 9: the same loop on 192 x 256 tiles (whole tiles only); 10: on 128 x 256 tiles (bf16 GEMMs)
    bool want_g8 = tile_req == 7 || tile_req == 8 || tile_req == 9 || (tile_req == 10 && a->dtype == MI355X_BF16 && !a->conv);
    if (tile_req == 11) {  // 192-row tiles for a whole number of rounds + 128-row tiles for a whole number of rounds (bf16 GEMMs whose shape admits it: plan_mix); else tile 9
        int rb, cb, nb, ns, dev = 0, ncu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        want_g8 = true;
        g8_mt = a->dtype == MI355X_BF16 && !a->conv && g_sk_g == 0 && plan_mix(p.M, p.N, ncu, rb, cb, nb, ns) ? 11 : 6;
    }
    if (want_g8 && p.ksplit > 1) {
        // a caller that split K for want of tiles AND asks for the 8-wave loop (native._fill_split: the measured table replaced a heuristic split): the loop needs
        // no split (whole tiles or stream-K) -- take it unsplit where it can run, otherwise keep the split on the 128 x 128 tile of the 4-wave kernel
        GemmP q = p;
        q.ksplit = 1;
        q.kb_per_split = 0;
        q.partial = nullptr;
        if (gemm8_ok(q, a->conv != 0, g8_mt)) p = q;
        else p.tile_hint = 1;
    }
    if (want_g8 && gemm8_ok(p, a->conv != 0, g8_mt)) {  // the 8-wave / eight-phase loop (gemm8_kernel.cuh); otherwise the heuristic decides
        ++g_stat_g8;
        if (p.lora_b) ++g_stat_g8_lora;
        if (g8_mt == 6) ++g_stat_g9;
        if (g8_mt == 11) ++g_stat_g11;
        const bool sk = tile_req == 8 && a->sk_ws && a->sk_flags && a->sk_slots > 0 && (reinterpret_cast<uintptr_t>(a->sk_ws) & 15) == 0;
        p.sk_ws = static_cast<float*>(a->sk_ws);
        p.sk_flags = a->sk_flags;
        p.sk_cap = a->sk_slots;
        if (a->conv) return a->dtype == MI355X_F32 ? launch_conv8_f32(p, st, sk, g8_mt) : launch_conv8_bf16(p, st, sk, g8_mt);
        return a->dtype == MI355X_F32 ? launch_gemm8_f32(p, st, sk, g8_mt) : launch_gemm8_bf16(p, st, sk, g8_mt);
    }
    if (a->conv) return a->dtype == MI355X_F32 ? launch_conv_f32(p, st) : launch_conv_bf16(p, st);
    if (a->dtype == MI355X_F32) return launch_tile<float, false>(p, st);
    return launch_tile<bf16_t, false>(p, st);
}
