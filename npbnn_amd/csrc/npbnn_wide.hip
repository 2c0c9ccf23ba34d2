// Host side of the weight-streamed path (device code and design: npbnn_wide.hip.h).  Not part of the ABI: the entry points of
// include/npbnn_hip.h pick this path by themselves (npbnn_set_arch: a layer of more than 128 nodes, or a weight image the LDS of a
// compute unit cannot hold with a useful number of waves beside it; NPBNN_OPT_WIDE) - the reference takes any shape
// (MatrixMultiplicationD, np_bnn/BNN_lib.py:154-162; default n_nodes = [50, 5] on any number of features, np_bnn/BNN_env.py:20).
#define NPBNN_KERNELS_WIDE
#include "npbnn_ctx.hip.h"

namespace npbnn_api {

namespace {

// tilings of wide_gemm_kernel: rows x outputs of a workgroup = 16 RT WR x 16 CT WC.  What a compute unit takes in by LDS-DMA grows
// with the number of waves that request it (tools/microbench_ingest.hip: 7 / 12 / 17 B per clock from the L2 with 4 / 8 / 16 waves,
// whatever each has in flight), and both operands come in that way: eight waves, and blocks as large as LDS and registers allow - the
// bytes a block reads per product fall with its size.
struct GemmCfg {
    wide_gemm_fn_t f16, f32;
    wide_gemm_fn_t f16_d2, f32_d2;   // the same tiling with two weight sets per pass (fused passes of a device chain), or nullptr
    wide_gemm_fn_t f16_d3, f32_d3;   // ... with three
    int xt, wt;            // row tiles / output tiles of a workgroup
    int threads, ppw;      // LDS-DMA pieces a wave requests per K-unit
    int n_stage;           // stages of the ring
    int wc;                // waves along the outputs (1: every wave holds whole rows of the block - the fused end applies)
    bool attr16 = false, attr32 = false, attr16_d2 = false, attr32_d2 = false, attr16_d3 = false, attr32_d3 = false;
};
GemmCfg g_cfg[6] = {
    {wide_gemm_kernel<8, 4, 2, 4, true>, wide_gemm_kernel<8, 4, 2, 4, false>, nullptr, nullptr, nullptr, nullptr, 16, 16, 512, 8, 2, 4},    // 256 x 256: 64 KiB per stage
    {wide_gemm_kernel<4, 4, 4, 2, true>, wide_gemm_kernel<4, 4, 4, 2, false>, nullptr, nullptr, nullptr, nullptr, 16, 8, 512, 6, 3, 2},     // 256 x 128: 48 KiB
    {wide_gemm_kernel<2, 4, 8, 1, true, 1, 3>, wide_gemm_kernel<2, 4, 8, 1, false, 1, 3>, wide_gemm_kernel<2, 4, 8, 1, true, 2, 3>, wide_gemm_kernel<2, 4, 8, 1, false, 2, 3>, wide_gemm_kernel<2, 4, 8, 1, true, 3, 3>, wide_gemm_kernel<2, 4, 8, 1, false, 3, 3>, 16, 4, 512, 5, 3, 1},     // 256 x 64 : 40 KiB
    {wide_gemm_kernel<2, 4, 4, 1, true, 1, 3>, wide_gemm_kernel<2, 4, 4, 1, false, 1, 3>, wide_gemm_kernel<2, 4, 4, 1, true, 2, 3>, wide_gemm_kernel<2, 4, 4, 1, false, 2, 3>, wide_gemm_kernel<2, 4, 4, 1, true, 3, 3>, wide_gemm_kernel<2, 4, 4, 1, false, 3, 3>, 8, 4, 256, 6, 3, 1},      // 128 x 64 : 24 KiB
    {wide_gemm_kernel<2, 2, 4, 1, true, 1, 3>, wide_gemm_kernel<2, 2, 4, 1, false, 1, 3>, wide_gemm_kernel<2, 2, 4, 1, true, 2, 3>, wide_gemm_kernel<2, 2, 4, 1, false, 2, 3>, wide_gemm_kernel<2, 2, 4, 1, true, 3, 3>, wide_gemm_kernel<2, 2, 4, 1, false, 3, 3>, 8, 2, 256, 5, 3, 1},      // 128 x 32 : 20 KiB
    {wide_gemm_kernel<4, 4, 4, 4, true>, wide_gemm_kernel<4, 4, 4, 4, false>, nullptr, nullptr, nullptr, nullptr, 16, 16, 1024, 4, 2, 4},   // 256 x 256 on 16 waves (experiment: NPBNN_WIDE_CFG=5)
};
// the tiling of a layer: by its width; tables of few rows take the 128-row blocks (more workgroups)
GemmCfg& cfg_for(int mt, int n_row_tiles, int n_cu) {
    if (const char* e = getenv("NPBNN_WIDE_CFG")) { const int v = atoi(e); if (v >= 0 && v < 6) return g_cfg[v]; }
    if (mt > 8) return g_cfg[0];
    if (mt > 4) return g_cfg[1];
    const bool few_rows = (n_row_tiles + 15) / 16 < n_cu / 2;
    if (mt > 2) return few_rows ? g_cfg[3] : g_cfg[2];
    return g_cfg[4];
}

// K-slices of a layer's product: while the blocks of the output do not fill the chip and a slice keeps a contraction worth its prologue
int slices_for(const GemmCfg& cf, int n_row_tiles, int mt, int units, int n_cu) {
    const int n_rb = (n_row_tiles + cf.xt - 1) / cf.xt, n_cb = (mt + cf.wt - 1) / cf.wt;
    int n_sl = 1;
    if (const char* e = getenv("NPBNN_WIDE_SLICES")) n_sl = atoi(e);
    else while (n_sl < kWideMaxSlices && n_rb * n_cb * (n_sl + 1) <= n_cu && units / (n_sl + 1) >= 8) ++n_sl;
    if (n_sl > kWideMaxSlices) n_sl = kWideMaxSlices;
    if (n_sl > units) n_sl = units;
    if (n_sl < 1) n_sl = 1;
    return n_sl;
}

// the narrow layers from layer `l` on as wide_tail_layers takes them, or false when they are not narrow (or too large for `lds_budget` bytes)
bool tail_desc(const npbnn_ctx* ctx, int l, const float* image, bool dev_slopes, size_t lds_budget, WideTailDesc* t, int* lds_floats) {
    const WideMeta& m = ctx->wmeta;
    *t = WideTailDesc{};
    int off = 0;
    for (int q = l; q < m.n_layers; ++q) {
        if (m.L[q].mt > kTailOut) return false;
        const int i = q - l;
        t->frag_off[i] = m.L[q].frag_off;
        t->bias_off[i] = m.L[q].bias_off;
        t->mt[i] = m.L[q].mt;
        t->frag_floats[i] = m.L[q].units * 2 * m.L[q].mt * 256;
        t->lds_frag[i] = off;
        off += t->frag_floats[i];
        t->lds_bias[i] = off;
        off += 16 * m.L[q].mt;
        t->act_prm[i] = ctx->net.act_prm[q];
    }
    if ((size_t)off * 4 > lds_budget) return false;
    t->n_layers = m.n_layers - l;
    t->last_is_output = 1;
    t->act_kind = ctx->net.act_kind;
    t->image = image;
    t->act_prm_dev = (dev_slopes && t->n_layers > 0) ? &ctx->d_slopes->cand[0][0][l] : nullptr;
    *lds_floats = off;
    return true;
}

// Does the first layer's product take the rest of the pass along (WideGemmArgs::fuse)?  Its tiling must give every wave whole rows (one
// output block, WC = 1), the contraction one K-slice, the remaining layers be narrow and fit the LDS beside the rows' scratch.
bool fused_pass(const npbnn_ctx* ctx, const Dataset& d, int* n_row_blocks, size_t* lds_need) {
    static const bool off = getenv("NPBNN_WIDE_NO_FUSE") != nullptr || getenv("NPBNN_WIDE_NO_TAIL") != nullptr;
    if (off) return false;
    const WideMeta& m = ctx->wmeta;
    const GemmCfg& cf = cfg_for(m.L[0].mt, d.n_tiles, ctx->n_cu);
    if (cf.wc != 1 || m.L[0].mt > cf.wt || m.L[0].mt > kTailIn) return false;
    if (slices_for(cf, d.n_tiles, m.L[0].mt, m.L[0].units, ctx->n_cu) != 1) return false;
    if (m.n_out > 128) return false;
    const int lik = ctx->net.lik_kind;          // (the float64 row-wise likelihoods and wide Gaussian targets stay with wide_lik_kernel)
    if (lik_needs_row_scratch(lik) || (lik == NPBNN_LIK_GAUSS && ctx->net.k_targets > kFuseTargets)) return false;
    WideTailDesc t;
    int tail_floats = 0;
    if (!tail_desc(ctx, 1, nullptr, false, 64 * 1024, &t, &tail_floats)) return false;
    const int ldz = ((m.n_out + 15) & ~15) + 1;
    const size_t need = ((size_t)kWideMaxCand * ((tail_floats + 3) & ~3) + (size_t)(cf.threads / 64) * 16 * ldz) * 4      // (room for two candidates' tails,
                        + 16 + (size_t)(cf.threads / 64) * kPartialStride * 8;                                              //  the rows' scratch, the waves' sums)
    if (need > 150 * 1024) return false;
    *n_row_blocks = (d.n_tiles + cf.xt - 1) / cf.xt;
    *lds_need = need;
    return true;
}

int stages_env() {
    static const int v = getenv("NPBNN_WIDE_STAGES") ? atoi(getenv("NPBNN_WIDE_STAGES")) : 0;
    return v;
}

// bytes of LDS the resident path would need for this network with one candidate and `waves` waves (build_net's layout arithmetic)
size_t resident_lds_bytes(const npbnn_ctx* ctx, const npbnn_arch* a, bool f16, int waves) {
    long long off = 0;
    int in = a->in_dim;
    bool narrow_later = a->n_layers >= 2;
    for (int l = 1; l < a->n_layers; ++l) narrow_later = narrow_later && a->out_dim[l] <= 16;
    const bool l1_f16 = f16 && narrow_later && a->act_kind == NPBNN_ACT_TANH && a->out_dim[0] > 16 && !a->final_act;
    for (int l = 0; l < a->n_layers; ++l) {
        const int out = a->out_dim[l], mt = (out + 15) / 16;
        if (l == 0) {
            const int units = f16 ? (in + 31) / 32 : (in + 15) / 16;
            int rows = 16;
            if (f16 && ctx->l0_blocks.empty() && mt >= 3 && out % 16 != 0) rows = (out + mt - 1) / mt;
            long long slots = (long long)mt * units;
            if (!ctx->l0_blocks.empty()) {       // block-structured first layer (npbnn_set_layer_mask): per output tile the hull of its K-units
                const int g16 = (in + 15) / 16, per_unit = f16 ? 2 : 1;
                slots = 0;
                for (int t = 0; t < mt; ++t) {
                    int b = units, e = 0;
                    for (int g = 0; g < g16; ++g)
                        if (ctx->l0_blocks[(size_t)t * g16 + g]) { const int u = g / per_unit; if (u < b) b = u; if (u + 1 > e) e = u + 1; }
                    if (e > b) slots += e - b;
                }
            }
            off += slots * (f16 ? 32 * rows : 256);
        } else if (l == 1 && l1_f16) {
            off += ((a->out_dim[0] + 15) / 16 + 1) / 2 * 512;
        } else {
            off += (long long)((in + 15) / 16) * mt * 256;
        }
        off += 16 * mt;
        in = out;
    }
    if (ctx->n_classw > 0) off += kResidentMaxWidth;
    off += kMaxLayers + 64;
    const int kt0 = f16 ? 2 * ((a->in_dim + 31) / 32) : (a->in_dim + 15) / 16;
    const WaveLayout lay = make_wave_layout(true, true, a->n_targets, kt0, a->lik_kind);
    return (size_t)off * 4 + (size_t)waves * lay.wave_lds + 64;
}

}  // namespace

// Does this network run on the weight-streamed path on this layer-0 layout?  A layer the resident builds have no tiles for, or an
// image that leaves the resident kernel fewer than kMinResidentWaves waves per compute unit (measured, tools/time_wide.py on
// 100k rows: wherever the resident kernel keeps a handful of waves it beats the streamed path - one launch per pass, several
// candidates per read of X, the persistent schedules - e.g. [32, 8] on 1024 features, 7 waves: 10.2 k against 6.6 k it/s;
// NPBNN_WIDE_MIN_WAVES: A/B switch).  Asked per layout: the float32 image of a first layer is larger than its fp16-split one
// (no compact rows), so a network may run resident on the fp16-split path and streamed on the float32 fallback.
bool wide_needed(const npbnn_ctx* ctx, const npbnn_arch* a, bool f16) {
    if (ctx->wide_option == 1) return true;
    if (const char* e = getenv("NPBNN_FORCE_WIDE")) { if (atoi(e) != 0) return true; }
    for (int l = 0; l < a->n_layers; ++l)
        if (a->out_dim[l] > kResidentMaxWidth) return true;
    static const int min_waves = getenv("NPBNN_WIDE_MIN_WAVES") ? atoi(getenv("NPBNN_WIDE_MIN_WAVES")) : kMinResidentWaves;
    // (never fewer than four: the resident kernel deals a workgroup's tiles to the four SIMDs of its compute unit and, there, to the
    // waves of that SIMD - with fewer than four waves the tiles of the SIMDs without one are never computed.  Until this round such
    // launches were planned whenever the image left room for 1-3 waves: sums over 1/4 to 3/4 of the rows, silently)
    int w = min_waves < 4 ? 4 : min_waves;
    // First layers of one or two output tiles on many rows leave the resident kernel EARLIER: with fewer than eight waves beside the
    // image it carries one candidate per pass at 4 TB/s and less, while the streamed path's 128 x 32 tiling (two workgroups per compute
    // unit, three candidates per fused pass) is at its best there - 100k rows, freshly initialised chains (tools/switch_sweep.py):
    // [32, 8] on 992 / 1024 features 10.5 / 10.3 k it/s resident (7 / 6 waves) against 14.8 / 13.6 k streamed; [16, 4] on 2048 features
    // 5.7 against 10.6 k; level at eight to ten waves.  (Chains that accept a third of their proposals or more keep about half of that
    // advantage.)  Few rows stay as they were: their passes are not fused, one candidate each.
    // (only where the streamed pass is the fused one that carries them: later layers of at most 128 nodes, categorical likelihood or
    // a Gaussian one of at most kFuseTargets columns, no class weights' / row-wise float64 terms - fused_pass)
    bool fusable = a->n_layers >= 2 && a->out_dim[a->n_layers - 1] <= 128 && !lik_needs_row_scratch(a->lik_kind) &&
                   (a->lik_kind == NPBNN_LIK_CATEGORICAL || (a->lik_kind == NPBNN_LIK_GAUSS && a->n_targets <= kFuseTargets));
    for (int l = 1; l < a->n_layers; ++l) fusable = fusable && a->out_dim[l] <= 128;
    if (!getenv("NPBNN_WIDE_MIN_WAVES") && fusable && a->out_dim[0] <= 32 && ctx->ds[0].X != nullptr && ctx->ds[0].n_rows >= 65536 && w < 8) w = 8;
    return resident_lds_bytes(ctx, a, f16, w) > ctx->lds_limit;
}

void wide_free(npbnn_ctx* ctx) {
    if (ctx->d_wide_cand) { (void)hipFree(ctx->d_wide_cand); ctx->d_wide_cand = nullptr; }
    for (int i = 0; i < 3; ++i)
        if (ctx->d_wide_act[i]) { (void)hipFree(ctx->d_wide_act[i]); ctx->d_wide_act[i] = nullptr; }
    ctx->wide_act_cap = 0;
    if (ctx->d_wide_cs) { (void)hipFree(ctx->d_wide_cs); ctx->d_wide_cs = nullptr; }
    if (ctx->d_prep_terms) { (void)hipFree(ctx->d_prep_terms); ctx->d_prep_terms = nullptr; ctx->prep_cap = 0; }
}

// image layout, device images, packed weight -> image position map (called by rebuild_net after build_net has filled ctx->net)
int wide_build(npbnn_ctx* ctx, bool f16) {
    const npbnn_arch& a = ctx->arch;
    WideMeta m{};
    m.n_layers = a.n_layers;
    m.n_out = a.out_dim[a.n_layers - 1];
    long long off = 0;
    int in = a.in_dim, woff = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        WideLayer& L = m.L[l];
        L.in_dim = in;
        L.out_dim = a.out_dim[l];
        L.has_bias = a.has_bias[l] ? 1 : 0;
        L.w_off = woff;
        L.mt = (L.out_dim + 15) / 16;
        L.units = (in + 31) / 32;
        L.f16 = (l == 0 && f16) ? 1 : 0;
        L.frag_off = off;
        off += wide_frag_items(L) * 4;
        woff += L.out_dim * (in + L.has_bias);
        in = L.out_dim;
    }
    for (int l = 0; l < a.n_layers; ++l) {
        m.L[l].bias_off = off;
        off += 16 * m.L[l].mt;
    }
    m.classw_off = -1;
    if (ctx->n_classw > 0) {
        m.classw_off = off;
        off += 16 * m.L[a.n_layers - 1].mt;
    }
    m.image_floats = (off + 63) / 64 * 64;
    // positions of fp16-split entries are 30-bit indices of 2-byte halves (ChainParams::pos): images of up to 2 GiB
    if (m.image_floats >= (1ll << 29))
        return fail(ctx, NPBNN_E_ARG, "network too large: a weight image of %.1f GiB (this backend holds up to 2 GiB)", (double)m.image_floats * 4 / (1 << 30) / 1.0);
    ctx->wmeta = m;
    wide_free(ctx);
    HIP_TRY(ctx, hipMalloc(&ctx->d_image, (size_t)m.image_floats * sizeof(float)));
    HIP_TRY(ctx, hipMemset(ctx->d_image, 0, (size_t)m.image_floats * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_wide_cand, (size_t)kWideMaxCand * m.image_floats * sizeof(float)));      // (a candidate image per weight set of a pass)
    HIP_TRY(ctx, hipMemset(ctx->d_wide_cand, 0, (size_t)kWideMaxCand * m.image_floats * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_wide_cs, sizeof(WideCandState)));
    HIP_TRY(ctx, hipMemset(ctx->d_wide_cs, 0, sizeof(WideCandState)));
    std::vector<int> map((size_t)ctx->n_weights);
    std::vector<float> scale, wscale;
    if (f16) {
        scale.assign((size_t)ctx->n_weights, 1.0f);
        wscale.resize((size_t)round_up(a.in_dim, 32));
        HIP_TRY(ctx, hipMemcpy(wscale.data(), ctx->d_wscale, wscale.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (int l = 0; l < m.n_layers; ++l) {
        const WideLayer& L = m.L[l];
        const int ld = L.in_dim + L.has_bias;
        for (int o = 0; o < L.out_dim; ++o) {
            const int mt = o / 16, u16 = o % 16;
            for (int j = 0; j < ld; ++j) {
                const size_t wi = (size_t)L.w_off + (size_t)o * ld + j;
                if (L.has_bias && j == 0) { map[wi] = (int)(L.bias_off + o); continue; }
                const int c = j - L.has_bias;
                if (L.f16) {
                    const int u = c / 32, kg = (c % 32) / 8, jj = c % 8;
                    const long long half_index = 2 * L.frag_off + ((((long long)u * L.mt + mt) * 2) * 64 + kg * 16 + u16) * 8 + jj;
                    map[wi] = (int)(0x80000000u | (unsigned)half_index);
                    scale[wi] = wscale[(size_t)c];
                } else {
                    const int kt = c / 16, kq = (c % 16) / 4, s = c % 4;
                    map[wi] = (int)(L.frag_off + (((long long)kt * L.mt + mt) * 64 + kq * 16 + u16) * 4 + s);
                }
            }
        }
    }
    HIP_TRY(ctx, hipMalloc(&ctx->d_w2img, map.size() * sizeof(int)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_w2img, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    if (f16) {
        HIP_TRY(ctx, hipMalloc(&ctx->d_w2scale, scale.size() * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_w2scale, scale.data(), scale.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return NPBNN_OK;
}

int wide_plan(npbnn_ctx* ctx, int which, LaunchPlan* lp, int want_cand) {
    Dataset& d = ctx->ds[which];
    const WideMeta& m = ctx->wmeta;
    int max_ld = 16;
    for (int l = 0; l < m.n_layers; ++l)
        if (16 * m.L[l].mt > max_ld) max_ld = 16 * m.L[l].mt;
    const size_t need = (size_t)d.n_tiles * 16 * (size_t)max_ld;
    if (need > ctx->wide_act_cap) {
        for (int i = 0; i < 3; ++i)
            if (ctx->d_wide_act[i]) { (void)hipFree(ctx->d_wide_act[i]); ctx->d_wide_act[i] = nullptr; }
        ctx->wide_act_cap = 0;
        for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipMalloc(&ctx->d_wide_act[i], need * sizeof(float)));
        int max_sl = 1;           // the K-slices' sums of a layer: room for the most any layer of this network is cut into on this table
        for (int l = 0; l < m.n_layers; ++l) {
            const int sl = slices_for(cfg_for(m.L[l].mt, d.n_tiles, ctx->n_cu), d.n_tiles, m.L[l].mt, m.L[l].units, ctx->n_cu);
            if (sl > max_sl) max_sl = sl;
        }
        if (max_sl > 1) HIP_TRY(ctx, hipMalloc(&ctx->d_wide_act[2], need * sizeof(float) * max_sl));
        ctx->wide_act_cap = need;
    }
    if (m.L[0].f16 && !d.X16w) {        // the fp16-split copy in piece order
        Dataset* home = &d;
        if (d.borrowed && ctx->data_owner) {
            npbnn_ctx* root = ctx->data_owner;
            while (root->data_owner) root = root->data_owner;
            home = &root->ds[which];
        }
        if (!home->X16w) {
            const int n_units = (d.F + 31) / 32;
            const size_t n_pad = (size_t)d.n_tiles * 16;
            HIP_TRY(ctx, hipMalloc(&home->X16w, n_pad * (size_t)n_units * 32 * sizeof(float)));
            const long long items = (long long)n_pad * n_units * 4;
            hipLaunchKernelGGL(split_x_tiled_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)d.X, (long long)n_pad, d.Fp,
                               n_units, (const float*)ctx->d_xscale, home->X16w);
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (home != &d) { d.X16w = home->X16w; d.x16w_borrowed = true; }
    }
    lp->fn = nullptr;
    lp->fn_spec = nullptr;
    lp->n_cand = 1;            // (two for fused passes that asked for them: below)
    lp->wpb = 4;
    lp->lds = 0;
    lp->fast = false;
    lp->wide = true;
    int grid = (int)((d.n_rows + 255) / 256);
    if (grid < 1) grid = 1;
    int n_rb = 0;
    size_t fuse_lds = 0;
    if (fused_pass(ctx, d, &n_rb, &fuse_lds)) {
        grid = n_rb;        // (one partial record per row block of the first layer's product)
        // two candidates per pass where the fused product has a build for them: its intake of X, not its arithmetic, bounds it
        static const bool one = getenv("NPBNN_WIDE_ONE_CAND") != nullptr;
        const GemmCfg& cf = cfg_for(m.L[0].mt, d.n_tiles, ctx->n_cu);
        if (want_cand >= 2 && !one && cf.f16_d2 != nullptr && !ctx->slopes_option) lp->n_cand = 2;
        static const int cap = getenv("NPBNN_WIDE_MAX_CAND") ? atoi(getenv("NPBNN_WIDE_MAX_CAND")) : kWideMaxCand;
        if (want_cand >= 3 && !one && cf.f16_d3 != nullptr && !ctx->slopes_option && cap >= 3) lp->n_cand = 3;
    }
    lp->grid = grid;             // workgroups that write a partial record each (wide_lik_kernel's, or the fused product's row blocks)
    lp->n_waves = grid;
    return NPBNN_OK;
}

void wide_pack(npbnn_ctx* ctx, const double* d_w, const double* d_col_override, float* image, int* flags) {
    const long long total = wide_item_count(ctx->wmeta);
    hipLaunchKernelGGL(wide_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_w, d_col_override,
                       ctx->n_classw ? ctx->d_classw : nullptr, image, ctx->wmeta, ctx->wmeta.L[0].f16 ? ctx->d_wscale : nullptr, flags);
}

int wide_forward(npbnn_ctx* ctx, int which, const float* image, bool chain_pass, bool only_layer0, int* info, int n_cand) {
    Dataset& d = ctx->ds[which];
    const WideMeta& m = ctx->wmeta;
    hipStream_t st = ctx->stream;
    const bool f16 = m.L[0].f16 != 0;
    const float* A = f16 ? d.X16w : d.X;
    long long lda = f16 ? 32ll * m.L[0].units : d.Fp;
    if (f16 && !d.X16w) return fail(ctx, NPBNN_E_STATE, "the weight-streamed path has no fp16-split copy of this matrix (internal error)");
    const PassDesc* pass = chain_pass ? reinterpret_cast<const PassDesc*>(reinterpret_cast<const char*>(ctx->d_eparams) + offsetof(EvalParams, pass_desc)) : nullptr;
    const bool dev_slopes = chain_pass && ctx->batch_slopes && ctx->d_slopes;
    static const bool no_tail = getenv("NPBNN_WIDE_NO_TAIL") != nullptr;
    static bool tail_attr = false;
    for (int l = 0; l < m.n_layers; ++l) {
        const WideLayer& L = m.L[l];
        // the narrow end of the network in one launch (wide_tail_kernel): every remaining layer of <= 128 nodes, behind <= 256 inputs,
        // their weights within 96 KiB of LDS
        if (l >= 1 && !no_tail && lda <= 16 * kTailIn) {
            WideTailArgs t{};
            int tail_floats = 0;
            if (tail_desc(ctx, l, image, dev_slopes, 96 * 1024, &t.t, &tail_floats)) {
                t.A = A;
                t.lda = lda;
                t.n_row_tiles = d.n_tiles;
                t.kt0 = (int)(lda / 16);
                t.out = ctx->d_wide_act[l & 1];
                t.ldo = 16 * m.L[m.n_layers - 1].mt;
                t.pass = pass;
                if (!tail_attr) {
                    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(wide_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024 + 1024));
                    tail_attr = true;
                }
                int grid = (d.n_tiles + 7) / 8;
                if (grid > 2 * ctx->n_cu) grid = 2 * ctx->n_cu;
                hipLaunchKernelGGL(wide_tail_kernel, dim3(grid), dim3(512), (size_t)tail_floats * 4, st, t);
                A = t.out;
                lda = t.ldo;
                break;
            }
        }
        GemmCfg& cf = cfg_for(L.mt, d.n_tiles, ctx->n_cu);
        WideGemmArgs g{};
        g.A = A;
        g.lda = lda;
        g.n_row_tiles = d.n_tiles;
        g.n_units = L.units;
        g.a_half_last = lda < 32ll * L.units ? 1 : 0;
        g.mt_total = L.mt;
        g.W = image + L.frag_off;
        g.bias = image + L.bias_off;
        g.out = ctx->d_wide_act[l & 1];
        g.ldo = 16 * L.mt;
        g.act_kind = l + 1 < m.n_layers ? ctx->net.act_kind : -1;
        g.act_prm = ctx->net.act_prm[l];
        g.act_prm_dev = (dev_slopes && l + 1 < m.n_layers) ? &ctx->d_slopes->cand[0][0][l] : nullptr;
        g.pass = pass;
        g.a_tiled = (l == 0 && f16) ? 1 : 0;
        int fuse_rb = 0;
        size_t fuse_lds = 0;
        const bool fuse = l == 0 && !only_layer0 && fused_pass(ctx, d, &fuse_rb, &fuse_lds);
        if (fuse) {
            int tail_floats = 0;
            (void)tail_desc(ctx, 1, image, dev_slopes, 64 * 1024, &g.tail, &tail_floats);
            g.fuse = 1;
            g.n_row_blocks = fuse_rb;
            g.p = ctx->d_eparams;
            g.image = image;
            g.classw_off = m.classw_off;
            if (m.n_layers == 1) g.act_kind = -1;
            g.cand_stride = m.image_floats;
        } else if (n_cand > 1) {
            return fail(ctx, NPBNN_E_STATE, "the weight-streamed path carries several candidates in fused passes only (internal error)");
        }
        int n_stage = stages_env() >= 2 ? stages_env() : cf.n_stage;
        const int n_sets = (l == 0 && !only_layer0 && n_cand > 1) ? n_cand : 1;
        const int stage_bytes = (cf.xt + n_sets * cf.wt) * 2048;
        const int ppw = 2 * (cf.xt + n_sets * cf.wt) / (cf.threads / 64);
        while (n_stage > 2 && ((size_t)n_stage * stage_bytes > ctx->lds_limit || (n_stage - 2) * ppw > kWideMaxYounger)) --n_stage;
        if (n_stage > L.units + 1) n_stage = L.units + 1 < 2 ? 2 : L.units + 1;
        // The four-wave tilings live on TWO workgroups per compute unit (four waves alone take in 7 B per clock, eight 12:
        // tools/microbench_ingest.hip): a third stage that costs the second workgroup its place costs more than it hides (128 x 32,
        // three candidates, 100k x 1280: 282 us with three stages and one workgroup per compute unit, 171 us with two and two).
        if (cf.threads == 256 && n_stage == 3 && stages_env() < 2) {
            const size_t lds3 = std::max((size_t)3 * stage_bytes, fuse ? fuse_lds : (size_t)0), lds2 = std::max((size_t)2 * stage_bytes, fuse ? fuse_lds : (size_t)0);
            if (ctx->lds_limit / lds3 < 2 && ctx->lds_limit / lds2 >= 2) n_stage = 2;
        }
        g.n_stage = n_stage;
        size_t lds = (size_t)n_stage * stage_bytes;
        if (fuse && fuse_lds > lds) lds = fuse_lds;
        const bool use16 = L.f16 != 0;
        const int sets = fuse ? n_cand : 1;
        wide_gemm_fn_t fn = sets == 3 ? (use16 ? cf.f16_d3 : cf.f32_d3) : sets == 2 ? (use16 ? cf.f16_d2 : cf.f32_d2) : (use16 ? cf.f16 : cf.f32);
        bool& attr = sets == 3 ? (use16 ? cf.attr16_d3 : cf.attr32_d3) : sets == 2 ? (use16 ? cf.attr16_d2 : cf.attr32_d2) : (use16 ? cf.attr16 : cf.attr32);
        if (fn == nullptr) return fail(ctx, NPBNN_E_STATE, "no build of the weight-streamed product for this launch (internal error)");
        if (!attr) {         // (the largest ring any launch asks for: once per kernel)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_limit));
            attr = true;
        }
        const int n_rb = (d.n_tiles + cf.xt - 1) / cf.xt, n_cb = (L.mt + cf.wt - 1) / cf.wt;
        int n_sl = slices_for(cf, d.n_tiles, L.mt, L.units, ctx->n_cu);
        if (n_sl > 1 && !ctx->d_wide_act[2]) n_sl = 1;
        float* const layer_out = ctx->d_wide_act[l & 1];
        g.k_slices = n_sl;
        g.slice_stride = (long long)d.n_tiles * 16 * g.ldo;
        if (n_sl > 1) g.out = ctx->d_wide_act[2];
        const int grid = (n_rb + 7) / 8 * 8 * n_cb * n_sl;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(cf.threads), lds, st, g);
        if (n_sl > 1) {
            const long long n_vec4 = g.slice_stride / 4;
            hipLaunchKernelGGL(wide_reduce_kernel, dim3((unsigned)((n_vec4 + 255) / 256)), dim3(256), 0, st, (const float*)ctx->d_wide_act[2], g.slice_stride, n_sl,
                               layer_out, n_vec4, g.act_kind, g.act_prm, g.act_prm_dev, pass);
        }
        A = layer_out;
        lda = g.ldo;
        if (l == 0 && info) { info[0] = 16 * cf.xt; info[1] = 16 * cf.wt; info[2] = n_sl; info[3] = n_rb * n_cb * n_sl; }
        if (fuse) { HIP_TRY(ctx, hipGetLastError()); return NPBNN_OK; }       // (the rest of the pass went with the product)
        if (l == 0 && only_layer0) return NPBNN_OK;
    }
    WideLikArgs la{};
    la.p = ctx->d_eparams;
    la.z = A;
    la.ldz = lda;
    la.image = image;
    la.classw_off = m.classw_off;
    la.final_prm_dev = nullptr;
    const int grid = (int)((d.n_rows + 255) / 256) < 1 ? 1 : (int)((d.n_rows + 255) / 256);
    hipLaunchKernelGGL(wide_lik_kernel, dim3(grid), dim3(256), 0, st, la);
    HIP_TRY(ctx, hipGetLastError());
    return NPBNN_OK;
}

int wide_cand_begin(npbnn_ctx* ctx) {
    for (int j = 0; j < kWideMaxCand; ++j)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_wide_cand + (size_t)j * ctx->wmeta.image_floats, ctx->d_image, (size_t)ctx->wmeta.image_floats * sizeof(float),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_wide_cs, 0, sizeof(WideCandState), ctx->stream));
    return NPBNN_OK;
}

// before a chain pass whose proposals are too wide for the step to keep the candidate image itself (chain_prepare: ChainParams::cand_image unset)
// make_them: the step left the making of the candidates to this launch (ChainParams::prep_terms, wide_cand_prepare_kernel)
void wide_cand_sync(npbnn_ctx* ctx, int M, int n_cand, bool make_them) {
    const unsigned grid = (unsigned)((M + 255) / 256);
    const long long stride = ctx->wmeta.image_floats;
    hipLaunchKernelGGL(wide_cand_restore_kernel, dim3(grid, n_cand), dim3(256), 0, ctx->stream, (const ChainParams*)ctx->d_cparams, (const WideCandState*)ctx->d_wide_cs,
                       ctx->d_wide_cand, (const float*)ctx->d_image, stride);
    if (make_them)
        hipLaunchKernelGGL(wide_cand_prepare_kernel, dim3(grid, n_cand), dim3(256), 0, ctx->stream, (const ChainParams*)ctx->d_cparams, ctx->d_wide_cs, ctx->d_wide_cand, stride);
    else
        hipLaunchKernelGGL(wide_cand_apply_kernel, dim3(grid, n_cand), dim3(256), 0, ctx->stream, (const ChainParams*)ctx->d_cparams, ctx->d_wide_cs, ctx->d_wide_cand, stride);
}

}  // namespace npbnn_api
