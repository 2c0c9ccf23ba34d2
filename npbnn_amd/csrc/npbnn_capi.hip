// C ABI of the gfx950 backend (see include/npbnn_hip.h for the contract and the reference
// functions each entry point replaces).  Host-side orchestration only: device buffers, launches,
// the HIP stream of the chain.  No torch, no BLAS library, no CPU fallback for the numerics.
#include <algorithm>
#define NPBNN_KERNELS_MAIN
#include "npbnn_ctx.hip.h"

namespace npbnn_api {

thread_local std::string g_last_error;

int fail(npbnn_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

void free_dataset(Dataset& d) {
    if (d.X && !d.borrowed) (void)hipFree(d.X);
    if (d.labels) (void)hipFree(d.labels);
    if (d.targets) (void)hipFree(d.targets);
    if (d.inst_w) (void)hipFree(d.inst_w);
    if (d.X16 && !d.borrowed) (void)hipFree(d.X16);
    if (d.X16w && !d.x16w_borrowed) (void)hipFree(d.X16w);
    d = Dataset();
}

void destroy_ctx(npbnn_ctx* c);

// a borrower lets go of its owner's matrices (before it uploads its own, or when it is destroyed)
void unshare_data(npbnn_ctx* ctx) {
    npbnn_ctx* owner = ctx->data_owner;
    if (!owner) return;
    for (int w = 0; w < 2; ++w)
        if (ctx->ds[w].borrowed) {
            ctx->ds[w].X = nullptr; ctx->ds[w].X16 = nullptr; ctx->ds[w].borrowed = false; ctx->ds[w].f16_state = 0;
            if (ctx->ds[w].x16w_borrowed) { ctx->ds[w].X16w = nullptr; ctx->ds[w].x16w_borrowed = false; }
        }
    ctx->d_xscale = nullptr;          // (the scales travel with the training matrix)
    ctx->d_wscale = nullptr;
    ctx->scale_F = 0;
    ctx->data_owner = nullptr;
    if (--owner->n_borrowers == 0 && owner->zombie) destroy_ctx(owner);
}

template <typename T>
int upload_matrix(npbnn_ctx* ctx, const T* X, int64_t n_rows, int32_t F, int which) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!X || n_rows <= 0 || F <= 0 || (which != 0 && which != 1))
        return fail(ctx, NPBNN_E_ARG, "set_data: bad arguments (rows=%lld, features=%d, which=%d)",
                    (long long)n_rows, F, which);
    if (n_rows > (int64_t)1 << 30) return fail(ctx, NPBNN_E_ARG, "set_data: too many rows");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->n_borrowers > 0) return fail(ctx, NPBNN_E_STATE, "set_data: %d other context(s) use this one's matrices (npbnn_share_data)", ctx->n_borrowers);
    if (ctx->data_owner) {       // a borrower that gets data of its own: all borrowed matrices go back first
        unshare_data(ctx);
        free_dataset(ctx->ds[0]);
        free_dataset(ctx->ds[1]);
    }
    Dataset& d = ctx->ds[which];
    free_dataset(d);
    if (which == 0) {            // the fp16-split scales come from the training matrix
        if (ctx->d_xscale) { (void)hipFree(ctx->d_xscale); ctx->d_xscale = nullptr; }
        if (ctx->d_wscale) { (void)hipFree(ctx->d_wscale); ctx->d_wscale = nullptr; }
        ctx->scale_F = 0;
        if (ctx->ds[1].X16) { (void)hipFree(ctx->ds[1].X16); ctx->ds[1].X16 = nullptr; }
        if (ctx->ds[1].X16w) { (void)hipFree(ctx->ds[1].X16w); ctx->ds[1].X16w = nullptr; }
        ctx->ds[1].f16_state = 0;
    }
    d.n_rows = n_rows;
    d.F = F;
    d.Fp = round_up(F, 16);
    d.n_tiles = (int)((n_rows + 15) / 16);
    const size_t n_pad = (size_t)d.n_tiles * 16;
    const size_t bytes = n_pad * d.Fp * sizeof(float);
    HIP_TRY(ctx, hipMalloc(&d.X, bytes));
    // convert + pad on the host in slabs, so the staging buffer stays small
    const size_t slab_rows = 16384;
    std::vector<float> stage(slab_rows * d.Fp);
    for (size_t r0 = 0; r0 < n_pad; r0 += slab_rows) {
        const size_t nr = std::min(slab_rows, n_pad - r0);
        std::fill(stage.begin(), stage.begin() + nr * d.Fp, 0.0f);
        for (size_t r = 0; r < nr; ++r) {
            const size_t gr = r0 + r;
            if (gr >= (size_t)n_rows) break;
            const T* src = X + gr * F;
            float* dst = stage.data() + r * d.Fp;
            for (int c = 0; c < F; ++c) dst[c] = (float)src[c];
        }
        HIP_TRY(ctx, hipMemcpy(d.X + r0 * d.Fp, stage.data(), nr * d.Fp * sizeof(float), hipMemcpyHostToDevice));
    }
    return NPBNN_OK;
}

int check_rows(npbnn_ctx* ctx, int which, int64_t n_rows, const char* what) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (which != 0 && which != 1) return fail(ctx, NPBNN_E_ARG, "%s: which must be 0 or 1", what);
    if (!ctx->ds[which].X) return fail(ctx, NPBNN_E_STATE, "%s: call npbnn_set_data first", what);
    if (ctx->ds[which].n_rows != n_rows)
        return fail(ctx, NPBNN_E_ARG, "%s: %lld rows but the data matrix has %lld", what, (long long)n_rows,
                    (long long)ctx->ds[which].n_rows);
    return NPBNN_OK;
}

int build_net(npbnn_ctx* ctx, const npbnn_arch* a, bool f16) {
    if (a->n_layers < 1 || a->n_layers > NPBNN_MAX_LAYERS)
        return fail(ctx, NPBNN_E_ARG, "set_arch: n_layers=%d outside 1..%d", a->n_layers, NPBNN_MAX_LAYERS);
    if (a->in_dim < 1) return fail(ctx, NPBNN_E_ARG, "set_arch: in_dim=%d", a->in_dim);
    if (a->act_kind < 0 || a->act_kind > NPBNN_ACT_TANH) return fail(ctx, NPBNN_E_ARG, "set_arch: act_kind=%d", a->act_kind);
    if (a->out_kind < 0 || a->out_kind > NPBNN_OUT_SOFTPLUS_HALF) return fail(ctx, NPBNN_E_ARG, "set_arch: out_kind=%d", a->out_kind);
    if (a->lik_kind < 0 || a->lik_kind > NPBNN_LIK_NONE) return fail(ctx, NPBNN_E_ARG, "set_arch: lik_kind=%d", a->lik_kind);
    NetMeta net{};
    net.n_layers = a->n_layers;
    net.act_kind = a->act_kind;
    net.out_kind = a->out_kind;
    net.lik_kind = a->lik_kind;
    net.k_targets = a->n_targets;
    net.final_act = a->final_act ? 1 : 0;
    net.l0_f16 = f16 ? 1 : 0;
    // softmax / categorical heads: padding outputs are masked through their bias (see NetMeta::pad_masked)
    net.pad_masked = (!net.final_act && (a->out_kind == NPBNN_OUT_SOFTMAX || a->lik_kind == NPBNN_LIK_CATEGORICAL)) ? 1 : 0;
    if (getenv("NPBNN_NO_PAD_MASK")) net.pad_masked = 0;       // (A/B timing)
    {   // layer 1 on fp16-split products (NetMeta::l1_f16)
        bool narrow_later = a->n_layers >= 2;
        for (int l = 1; l < a->n_layers; ++l) narrow_later = narrow_later && a->out_dim[l] <= 16;
        net.l1_f16 = (f16 && narrow_later && a->act_kind == NPBNN_ACT_TANH && a->out_dim[0] > 16 && !a->final_act && !getenv("NPBNN_NO_L1_F16")) ? 1 : 0;
    }
    int in = a->in_dim, off = 0, woff = 0;
    for (int l = 0; l < a->n_layers; ++l)
        if (a->out_dim[l] < 1 || a->out_dim[l] > NPBNN_MAX_WIDTH)
            return fail(ctx, NPBNN_E_ARG, "set_arch: layer %d has %d nodes; this backend supports 1..%d per layer", l, a->out_dim[l],
                        NPBNN_MAX_WIDTH);
    {
        long long nw = 0;
        int in_l = a->in_dim;
        for (int l = 0; l < a->n_layers; ++l) { nw += (long long)a->out_dim[l] * (in_l + (a->has_bias[l] ? 1 : 0)); in_l = a->out_dim[l]; }
        if (nw >= (1ll << 31)) return fail(ctx, NPBNN_E_ARG, "set_arch: %lld weights; this backend indexes up to 2^31", nw);
    }
    // Networks the LDS of a compute unit cannot hold (or a layer wider than the resident builds' tiles) run on the weight-streamed
    // path (npbnn_wide.hip): the description below then carries what the chain step and the likelihood need - shapes, offsets into
    // the packed weights, kinds - and none of the resident image's layout.
    ctx->wide = wide_needed(ctx, a, f16);
    if (ctx->wide) {
        net.pad_masked = 0;
        net.l1_f16 = 0;
        net.l0_rows = 16;
        for (int mt = 0; mt < kMaxMT; ++mt) { net.l0_begin[mt] = 0; net.l0_end[mt] = 0; net.l0_base[mt] = 0; }
        for (int l = 0; l < a->n_layers; ++l) {
            LayerMeta& L = net.L[l];
            L.in_dim = in;
            L.out_dim = a->out_dim[l];
            L.has_bias = a->has_bias[l] ? 1 : 0;
            L.kt = (in + 15) / 16;
            L.mt = (L.out_dim + 15) / 16;
            L.frag_off = 0;
            L.bias_off = 0;
            L.out_perm = 0;
            L.in_live = 4;
            L.w_off = woff;
            woff += L.out_dim * (in + L.has_bias);
            in = L.out_dim;
        }
        net.classw_off = -1;
        net.slope_off = ctx->slopes_option ? 0 : -1;     // (>= 0 says "candidates carry their own slopes": the streamed kernels read them from the chain)
        net.image_floats = 0;
    }
    for (int l = 0; l < a->n_layers && !ctx->wide; ++l) {
        const int out = a->out_dim[l];
        LayerMeta& L = net.L[l];
        L.in_dim = in;
        L.out_dim = out;
        L.has_bias = a->has_bias[l] ? 1 : 0;
        L.kt = (l == 0 && f16) ? 2 * ((in + 31) / 32) : (in + 15) / 16;   // 1-KiB pieces (16 rows x 64 B) per tile
        L.mt = (out + 15) / 16;
        L.frag_off = off;
        L.out_perm = (l >= 1 && l + 1 < a->n_layers && L.mt == 1 && !getenv("NPBNN_NO_TILE_PERM")) ? 1 : 0;
        L.in_live = (l >= 1 && net.L[l - 1].out_perm) ? (net.L[l - 1].out_dim + 3) / 4 : 4;
        if (l == 0) {
            // K-units of the layer-0 loop (32 features on the fp16-split path, 16 on the float32 one); per output tile the hull of
            // the units in which the mask has anything (the whole layer when no structure was declared)
            const int g16 = (in + 15) / 16, per_unit = f16 ? 2 : 1, units = f16 ? (in + 31) / 32 : g16;
            int slots = 0;
            for (int mt = 0; mt < kMaxMT; ++mt) { net.l0_begin[mt] = 0; net.l0_end[mt] = 0; net.l0_base[mt] = 0; }
            for (int mt = 0; mt < L.mt; ++mt) {
                int b = 0, e = units;
                if (!ctx->l0_blocks.empty()) {
                    b = units; e = 0;
                    for (int g = 0; g < g16; ++g)
                        if (ctx->l0_blocks[(size_t)mt * g16 + g]) { const int u = g / per_unit; if (u < b) b = u; if (u + 1 > e) e = u + 1; }
                    if (e <= b) { b = 0; e = 0; }
                }
                net.l0_begin[mt] = b; net.l0_end[mt] = e; net.l0_base[mt] = slots;
                slots += e - b;
            }
            // rows per tile in the image (NetMeta::l0_rows): a dense fp16-split first layer of three or more tiles whose width is not a
            // multiple of 16 is stored without its padding rows (the builds for one and two tiles - every BASELINE shape - keep 16)
            net.l0_rows = 16;
            if (f16 && ctx->l0_blocks.empty() && L.mt >= 3 && out % 16 != 0 && !getenv("NPBNN_NO_COMPACT_ROWS")) net.l0_rows = (out + L.mt - 1) / L.mt;
            off += slots * (f16 ? 32 * net.l0_rows : 256);
        } else if (l == 1 && net.l1_f16) {
            off += ((net.L[0].mt + 1) / 2) * 512;      // a high and a low block of 256 floats per K-step
        } else {
            off += L.kt * L.mt * 256;
        }
        L.w_off = woff;
        woff += out * (in + L.has_bias);
        in = out;
    }
    if (!ctx->wide) {
    for (int l = 0; l < a->n_layers; ++l) {
        net.L[l].bias_off = off;
        off += 16 * net.L[l].mt;
    }
    if (ctx->n_classw > 0) {            // class weights ride in the image only when there are any
        net.classw_off = off;
        off += kResidentMaxWidth;
    } else {
        net.classw_off = -1;
    }
    net.slope_off = -1;
    if (ctx->slopes_option) {           // a slot per hidden layer for the candidates' activation slopes (filled in LDS by a chain pass)
        net.slope_off = off;
        off += kMaxLayers;
    }
    net.image_floats = round_up(off, 64);   // a multiple of 256 B (the LDS copies of several candidates sit back to back)
    }
    net.n_out = a->out_dim[a->n_layers - 1];
    if (a->lik_kind == NPBNN_LIK_GAUSS) {
        if (a->n_targets < 1 || a->n_targets > NPBNN_MAX_TARGETS || a->n_targets > net.n_out)
            return fail(ctx, NPBNN_E_ARG, "set_arch: Gaussian likelihood needs 1..%d target columns (<= outputs), got %d",
                        NPBNN_MAX_TARGETS, a->n_targets);
    } else if (lik_needs_row_scratch(a->lik_kind)) {
        const int k = a->n_targets;
        int need_out = 1;
        if (a->lik_kind == NPBNN_LIK_GAUSS_PRED_SIGMA || a->lik_kind == NPBNN_LIK_NEGBIN2D) need_out = 2 * k;
        else if (a->lik_kind != NPBNN_LIK_POISSON) need_out = 2;
        if (k < 1 || k > 8 || net.n_out > 16 || net.n_out < need_out)
            return fail(ctx, NPBNN_E_ARG, "set_arch: likelihood kind %d needs 1..8 target columns and %d..16 outputs (got %d targets, %d outputs)",
                        a->lik_kind, need_out, k, net.n_out);
    }
    ctx->net = net;
    ctx->n_weights = woff;
    ctx->mt0_template = net.L[0].mt;
    return NPBNN_OK;
}

// waves per block such that the fragment image + per-wave rings fit the CU's LDS
int max_inner_tiles(const NetMeta& net) {
    int mti = 1;
    for (int l = 1; l < net.n_layers; ++l)
        if (net.L[l].mt > mti) mti = net.L[l].mt;
    if (net.n_layers == 1) mti = net.L[0].mt;        // the single layer's tiles are also the final tiles
    return mti;
}

WaveLayout layout_for(const npbnn_ctx* ctx, const Dataset& d, bool predict_only) {
    return make_wave_layout(d.labels != nullptr, d.inst_w != nullptr, d.targets ? ctx->net.k_targets : 0, ctx->net.L[0].kt,
                            predict_only ? NPBNN_LIK_NONE : ctx->net.lik_kind);
}

int pick_waves_per_block(const npbnn_ctx* ctx, size_t* lds_bytes, int n_cand, const WaveLayout& lay, bool predict_only, bool fast) {
    const int lk = predict_only ? kLikCat : lik_class(ctx->net.lik_kind);
    const int top = max_waves_for(ctx->net.L[0].mt, max_inner_tiles(ctx->net) == 1 ? 1 : 8, ctx->net.l0_f16 != 0, n_cand, lk, fast);   // launch bound of the build in use
    for (int w = top; w >= 1; --w) {
        const size_t need = (size_t)n_cand * ctx->net.image_floats * 4 + (size_t)w * lay.wave_lds;
        if (need + 64 <= ctx->lds_limit) {       // (+ 64: the flag word of the device-side waits, plan_launch)
            *lds_bytes = need;
            return w;
        }
    }
    return 0;
}


eval_fn_t pick_kernel(const NetMeta& net, int n_cand) {
    return npbnn_pick_eval_kernel(net.L[0].mt, max_inner_tiles(net) == 1 ? 1 : 8, net.l0_f16, n_cand, lik_class(net.lik_kind));
}

// Per column of a resident matrix under the context's scales (split_quality_kernel): how far the fp16 pair's largest counted entry
// error is from its bounds - max(error / (2^-17 x mean |entry|), error / (2^-12 x typical |entry|)); <= 1 passes, 0 for an exact column.
static int column_quality(npbnn_ctx* ctx, const Dataset& d, std::vector<double>* badness) {
    const int Fq = d.Fp;
    unsigned* d_err = nullptr;
    unsigned long long* d_sum = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_err, (size_t)Fq * sizeof(unsigned)));
    HIP_TRY(ctx, hipMalloc(&d_sum, (size_t)Fq * 3 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(d_err, 0, (size_t)Fq * sizeof(unsigned), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_sum, 0, (size_t)Fq * 3 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(split_quality_kernel, dim3((Fq + 255) / 256, (unsigned)((d.n_rows + 1023) / 1024)), dim3(256), 0, ctx->stream,
                       (const float*)d.X, (long long)d.n_rows, d.Fp, (const float*)ctx->d_xscale, d_err, d_sum, d_sum + Fq, d_sum + 2 * (size_t)Fq);
    std::vector<unsigned> h_err((size_t)Fq);
    std::vector<unsigned long long> h_sum((size_t)Fq * 3);
    HIP_TRY(ctx, hipMemcpyAsync(h_err.data(), d_err, h_err.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h_sum.data(), d_sum, h_sum.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_err);
    (void)hipFree(d_sum);
    badness->assign((size_t)d.F, 0.0);
    for (int c = 0; c < d.F; ++c) {
        float e;
        memcpy(&e, &h_err[(size_t)c], 4);
        const unsigned long long cnt = h_sum[2 * (size_t)Fq + c];
        if (cnt == 0 || !(e > 0.f)) continue;                 // an all-zero column, or one the pair holds exactly
        const double mean_abs = (double)h_sum[(size_t)c] / 268435456.0 / (double)d.n_rows;
        const double typical = std::exp2((double)(long long)h_sum[(size_t)Fq + c] / 65536.0 / (double)cnt);
        const double by_mean = mean_abs > 0.0 ? (double)e / mean_abs / (double)kF16QualityTol : 1e300;
        const double by_typical = (double)e / typical / (double)kF16TypicalTol;
        (*badness)[(size_t)c] = by_mean > by_typical ? by_mean : by_typical;
    }
    return NPBNN_OK;
}

// ---- fp16-split data: scales from the training matrix, split copies built on the device ----
int ensure_scales(npbnn_ctx* ctx) {
    Dataset& tr = ctx->ds[0];
    if (!tr.X) return fail(ctx, NPBNN_E_STATE, "the fp16-split path needs the training matrix first");
    if (ctx->d_xscale && ctx->scale_F == tr.F) return NPBNN_OK;
    const int Fp16 = round_up(tr.F, 32);
    unsigned* d_max = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_max, (size_t)Fp16 * sizeof(unsigned)));
    HIP_TRY(ctx, hipMemsetAsync(d_max, 0, (size_t)Fp16 * sizeof(unsigned), ctx->stream));
    const int row_blocks = (int)((tr.n_rows + 1023) / 1024);
    hipLaunchKernelGGL(col_absmax_kernel, dim3((tr.Fp + 255) / 256, row_blocks), dim3(256), 0, ctx->stream, tr.X,
                       (long long)tr.n_rows, tr.Fp, d_max);
    if (!ctx->d_xscale) HIP_TRY(ctx, hipMalloc(&ctx->d_xscale, (size_t)Fp16 * sizeof(float)));
    if (!ctx->d_wscale) HIP_TRY(ctx, hipMalloc(&ctx->d_wscale, (size_t)Fp16 * sizeof(float)));
    hipLaunchKernelGGL(col_scale_kernel, dim3((Fp16 + 255) / 256), dim3(256), 0, ctx->stream, d_max, Fp16, ctx->d_xscale,
                       ctx->d_wscale, (const int*)nullptr);
    std::vector<unsigned> h((size_t)Fp16);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_max, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->scale_F = tr.F;
    ctx->f16_shifted_cols = 0;
    ctx->f16_max_shift = 0;
    tr.f16_state = 0;
    for (unsigned bits : h) {
        float m;
        memcpy(&m, &bits, 4);
        if (!std::isfinite(m)) tr.f16_state = -1;     // inf / NaN in the data: stay on the exact float32 path
    }
    // Heavy-tailed columns: with the largest entry just under 1 the typical entries sit where the pair's absolute error floor
    // (2^-25) is a visible fraction of them.  Such a column's scale moves up by the power of two that brings its error / typical |value|
    // under the bound with a factor 2 to spare (the fp16 range above 1 is otherwise unused); the weights' scale moves down with it.
    // Columns inside the bound keep the scale they always had.
    if (tr.f16_state == 0 && !getenv("NPBNN_F16_NO_SHIFT")) {
        std::vector<double> ratio;
        int rcq = column_quality(ctx, tr, &ratio);
        if (rcq) { (void)hipFree(d_max); return rcq; }
        std::vector<int> shift((size_t)Fp16, 0);
        for (int c = 0; c < tr.F; ++c) {
            if (!(ratio[(size_t)c] > 1.0)) continue;
            int k = (int)std::ceil(std::log2(ratio[(size_t)c])) + 1;
            if (k > kF16MaxShift) k = kF16MaxShift;
            shift[(size_t)c] = k;
            ++ctx->f16_shifted_cols;
            if (k > ctx->f16_max_shift) ctx->f16_max_shift = k;
        }
        if (ctx->f16_shifted_cols > 0) {
            int* d_shift = nullptr;
            HIP_TRY(ctx, hipMalloc(&d_shift, (size_t)Fp16 * sizeof(int)));
            HIP_TRY(ctx, hipMemcpy(d_shift, shift.data(), (size_t)Fp16 * sizeof(int), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(col_scale_kernel, dim3((Fp16 + 255) / 256), dim3(256), 0, ctx->stream, d_max, Fp16, ctx->d_xscale,
                               ctx->d_wscale, (const int*)d_shift);
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(d_shift);
        }
    }
    (void)hipFree(d_max);
    return NPBNN_OK;
}

// 1 = the set has a usable fp16-split copy, 0 = it cannot be represented (the caller stays on float32)
int ensure_x16(npbnn_ctx* ctx, int which, int* usable) {
    *usable = 0;
    if (ctx->data_owner && ctx->ds[which].borrowed) {      // borrowed matrices come with their split copy, or without one
        *usable = ctx->ds[which].f16_state > 0 ? 1 : 0;
        return NPBNN_OK;
    }
    int rc = ensure_scales(ctx);
    if (rc) return rc;
    if (ctx->ds[0].f16_state < 0) return NPBNN_OK;
    Dataset& d = ctx->ds[which];
    if (d.f16_state == 0) {
        d.Fp16 = round_up(d.F, 32);
        const size_t n_pad = (size_t)d.n_tiles * 16;
        if (!d.X16) HIP_TRY(ctx, hipMalloc(&d.X16, n_pad * d.Fp16 * sizeof(float)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(int), ctx->stream));
        const long long items = (long long)n_pad * (d.Fp16 / 8);
        hipLaunchKernelGGL(split_x_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream, d.X, (long long)n_pad, d.Fp,
                           d.Fp16, ctx->d_xscale, d.X16, reinterpret_cast<unsigned*>(ctx->d_overflow));
        unsigned bits = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&bits, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        float m;
        memcpy(&m, &bits, 4);
        d.f16_state = (std::isfinite(m) && m <= kF16Safe) ? 1 : -1;    // a test set far outside the training range
        if (d.f16_state > 0) {      // and is the pair of fp16 numbers a fair picture of every column? (split_quality_kernel)
            std::vector<double> ratio;
            int rcq = column_quality(ctx, d, &ratio);
            if (rcq) return rcq;
            d.f16_worst_col = -1;
            d.f16_worst_ratio = 0.0;
            for (int c = 0; c < d.F; ++c)
                if (ratio[(size_t)c] > d.f16_worst_ratio) { d.f16_worst_ratio = ratio[(size_t)c]; d.f16_worst_col = c; }
            if (d.f16_worst_ratio > 1.0 && !getenv("NPBNN_F16_NO_QUALITY_CHECK")) d.f16_state = -2;   // heavy-tailed column(s) past what a moved scale holds
        }
        if (d.f16_state < 0 && d.X16) { (void)hipFree(d.X16); d.X16 = nullptr; }      // (nobody will read it)
    }
    *usable = d.f16_state > 0 ? 1 : 0;
    return NPBNN_OK;
}

// May a launch that wants nothing but the likelihood terms run on the fast builds (eval_kernel, FAST)?  2 or 3 layers, later
// layers of <= 16 nodes, layer 0 of <= 16 * kFastMaxMT0, categorical (padding outputs masked through the bias) or Gaussian
// likelihood, no row or class weights, no activation after the last layer.
// does layer 0 skip anything (some output tile without weights in some K-unit)?
bool l0_blocked(const NetMeta& net) {
    const int units = net.l0_f16 ? net.L[0].kt / 2 : net.L[0].kt;
    for (int mt = 0; mt < net.L[0].mt; ++mt)
        if (net.l0_begin[mt] != 0 || net.l0_end[mt] != units) return true;
    return false;
}

bool fast_launch_ok(const npbnn_ctx* ctx, const Dataset& d) {
    const NetMeta& net = ctx->net;
    if (!ctx->fast_option || max_inner_tiles(net) != 1 || net.n_layers < 2 || net.n_layers > kFastLayers || net.L[0].mt > kFastMaxMT0) return false;
    if (net.final_act || d.inst_w || ctx->n_classw > 0 || net.slope_off >= 0) return false;
    if (l0_blocked(net) && !net.l0_f16) return false;          // (the fast builds for block-structured layers are fp16-split ones)
    if (net.lik_kind == NPBNN_LIK_CATEGORICAL) return net.pad_masked != 0 && d.labels != nullptr;
    return net.lik_kind == NPBNN_LIK_GAUSS && d.targets != nullptr && net.k_targets <= (l0_blocked(net) ? 1 : kFastGaussTargets);
}

// lik_only: the caller wants the likelihood terms and nothing else from the launch (no statistics, no predictions)
int plan_launch(npbnn_ctx* ctx, int which, LaunchPlan* lp, int force_f32, int want_cand, bool predict_only, bool lik_only, bool plain) {
    Dataset& d = ctx->ds[which];
    bool want_f16 = false;
    if (!force_f32 && ctx->l0_option != NPBNN_L0_F32) {
        int usable = 0;
        int rc0 = ensure_x16(ctx, which, &usable);
        if (rc0) return rc0;
        if (!usable && ctx->l0_option == NPBNN_L0_F16) {
            if (d.f16_state == -2)
                return fail(ctx, NPBNN_E_RANGE, "fp16-split layer 0 was requested but column %d spans too many powers of two for a pair of fp16 "
                                                "numbers, even with its scale moved as far as fp16 allows (largest entry error %.1f x the bound: 2^-17 of the column's mean, "
                                                "2^-12 of its typical |value|)", d.f16_worst_col, d.f16_worst_ratio);
            return fail(ctx, NPBNN_E_RANGE, "fp16-split layer 0 was requested but the data cannot be represented in it");
        }
        want_f16 = usable != 0;
    }
    if ((ctx->net.l0_f16 != 0) != want_f16) {
        int rc0 = rebuild_net(ctx, want_f16);
        if (rc0) return rc0;
    }
    if (ctx->wide) return wide_plan(ctx, which, lp, (predict_only || !lik_only) ? 1 : want_cand);
    lp->wide = false;
    size_t lds = 0;
    // speculative passes: as many candidates as still leave >= 8 waves per workgroup (only the MTI = 1 builds have them)
    int n_cand = (max_inner_tiles(ctx->net) == 1 && (predict_only || !lik_needs_row_scratch(ctx->net.lik_kind))) ? want_cand : 1;
    if (n_cand > kMaxCand) n_cand = kMaxCand;
    if (n_cand > max_cand_for(ctx->net.L[0].mt)) n_cand = max_cand_for(ctx->net.L[0].mt);
    const WaveLayout lay = layout_for(ctx, d, predict_only);
    const bool fast = lik_only && !predict_only && fast_launch_ok(ctx, d);
    while (n_cand > 1 && pick_waves_per_block(ctx, &lds, n_cand, lay, predict_only, fast) < 8) --n_cand;
    lp->n_cand = n_cand;
    lp->fast = fast;
    int wpb = pick_waves_per_block(ctx, &lds, n_cand, lay, predict_only, fast);
    if (wpb == 0)
        return fail(ctx, NPBNN_E_ARG, "network too large: weight image of %d KiB does not fit the %zu KiB LDS of a CU",
                    ctx->net.image_floats * 4 / 1024, ctx->lds_limit / 1024);
    lp->fn = predict_only ? npbnn_pick_eval_kernel(ctx->net.L[0].mt, max_inner_tiles(ctx->net) == 1 ? 1 : 8, ctx->net.l0_f16, n_cand, kLikCat)
                          : npbnn_pick_eval_kernel(ctx->net.L[0].mt, max_inner_tiles(ctx->net) == 1 ? 1 : 8, ctx->net.l0_f16, n_cand,
                                                   lik_class(ctx->net.lik_kind), fast, fast && l0_blocked(ctx->net), plain && fast);
    if (!lp->fn) return fail(ctx, NPBNN_E_STATE, "no evaluation kernel for this shape (internal error)");
    lp->fn_spec = nullptr;
    if (fast && !plain && !predict_only && !l0_blocked(ctx->net) && (n_cand == 1 || n_cand == 3)) {
        const bool g = lik_class(ctx->net.lik_kind) == kLikGauss;
        const int mt0 = ctx->net.L[0].mt, f16 = ctx->net.l0_f16;
        lp->fn_spec = n_cand == 1 ? (g ? pick_eval_d1_gauss_spec(mt0, f16) : pick_eval_d1_cat_spec(mt0, f16))
                                  : (g ? pick_eval_d3_gauss_spec(mt0, f16) : pick_eval_d3_cat_spec(mt0, f16));
    }
    {
        // A thin second round of tiles: when a workgroup's share is just over one tile per wave (config 5 with three candidates: 12.25
        // tiles for 12 waves) the leftover tile runs alone after the others are through - a lone wave streams X at a ring's worth per
        // memory latency.  Two waves fewer spread the leftovers over several SIMDs and give the first round less contention: measured
        // 36.4 -> 33.4 us per pass there (9 or 10 waves; 11: 35.0, 8: 35.0, 7: 41.9); single-candidate launches and shares of two rounds
        // or more are best at the build's full count (config 2: 30.6 us at 12 waves, 31.3 at 10).  NPBNN_WAVES: A/B switch.
        int w_use = wpb;
        const double share = (double)d.n_tiles / (double)(ctx->n_cu > 0 ? ctx->n_cu : 1);
        if (n_cand > 1 && share > (double)w_use && share < 1.25 * (double)w_use && w_use - 2 >= 8) w_use -= 2;
        if (const char* e = getenv("NPBNN_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= wpb) w_use = v; }
        if (w_use != wpb) {
            wpb = w_use;
            lds = (size_t)n_cand * ctx->net.image_floats * 4 + (size_t)wpb * lay.wave_lds;
        }
    }
    // (the tile schedule of eval_kernel needs a wave on every SIMD of the compute unit: networks that leave fewer run on the
    // weight-streamed path - wide_needed - and a data set whose row-aux slots push a launch below that is refused rather than mis-summed)
    if (wpb < 4)
        return fail(ctx, NPBNN_E_ARG, "network too large for the LDS-resident path on this data set (%d waves beside a weight image of %d KiB); "
                                      "NPBNN_OPT_WIDE = 1 runs it on the weight-streamed path", wpb, ctx->net.image_floats * 4 / 1024);
    lp->wpb = wpb;
    lp->lds = lds;
    int grid = (d.n_tiles + wpb - 1) / wpb;
    if (grid > ctx->n_cu) grid = ctx->n_cu;     // persistent: one workgroup per CU
    if (grid < 1) grid = 1;
    lp->grid = grid;
    lp->n_waves = grid;            // one partial record per workgroup
    lp->lds = lds + 64;            // (+ the flag word of the device-side waits, behind the images and the rings)
    if (ctx->attr_fn != reinterpret_cast<const void*>(lp->fn) || ctx->attr_lds < lp->lds) {      // (not free: once per kernel and size)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lp->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp->lds));
        if (lp->fn_spec)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lp->fn_spec), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp->lds));
        ctx->attr_fn = reinterpret_cast<const void*>(lp->fn);
        ctx->attr_lds = lp->lds;
    }
    return NPBNN_OK;
}

int ensure_work_buffers(npbnn_ctx* ctx, int n_waves) {
    if (n_waves > ctx->partial_waves) {
        if (ctx->d_partials) (void)hipFree(ctx->d_partials);
        ctx->d_partials = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->d_partials, (size_t)2 * kMaxCand * n_waves * kPartialStride * sizeof(double)));   // two pass parities
        ctx->partial_waves = n_waves;
    }
    return NPBNN_OK;
}

int stage_weights(npbnn_ctx* ctx, const double* W, const double* act_prm, const double* col_override) {
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "call npbnn_set_arch first");
    if (!W) return fail(ctx, NPBNN_E_ARG, "null weights");
    memcpy(ctx->h_w, W, (size_t)ctx->n_weights * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_wraw, ctx->h_w, (size_t)ctx->n_weights * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const double* d_co = nullptr;
    if (col_override) {
        double* h_co = ctx->h_w + ctx->n_weights;
        memcpy(h_co, col_override, (size_t)ctx->arch.in_dim * sizeof(double));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_colov, h_co, (size_t)ctx->arch.in_dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        d_co = ctx->d_colov;
    }
    for (int l = 0; l < kMaxLayers; ++l) ctx->net.act_prm[l] = 0.f;
    if (act_prm)
        for (int l = 0; l + 1 < ctx->net.n_layers; ++l) ctx->net.act_prm[l] = (float)act_prm[l];
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(int), ctx->stream));
    launch_pack_weights(ctx, ctx->d_wraw, d_co, ctx->d_image, ctx->d_overflow);
    HIP_TRY(ctx, hipGetLastError());
    return NPBNN_OK;
}

// copy a parameter block to its device slot through the pinned staging area (stream ordered; the staging slot is
// reused only after the stream has been synchronised by the caller's epilogue)
int push_eval_params(npbnn_ctx* ctx, const EvalParams& p) {
    memcpy(ctx->h_params, &p, sizeof(EvalParams));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eparams, ctx->h_params, sizeof(EvalParams), hipMemcpyHostToDevice, ctx->stream));
    return NPBNN_OK;
}
int push_finalize_params(npbnn_ctx* ctx, const FinalizeParams& f) {
    char* slot = ctx->h_params + sizeof(EvalParams);
    memcpy(slot, &f, sizeof(FinalizeParams));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_fparams, slot, sizeof(FinalizeParams), hipMemcpyHostToDevice, ctx->stream));
    return NPBNN_OK;
}
int push_chain_params(npbnn_ctx* ctx, const ChainParams& c) {
    char* slot = ctx->h_params + sizeof(EvalParams) + sizeof(FinalizeParams);
    memcpy(slot, &c, sizeof(ChainParams));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_cparams, slot, sizeof(ChainParams), hipMemcpyHostToDevice, ctx->stream));
    return NPBNN_OK;
}

// diagnostics (NPBNN_EVAL_STAMPS): what the evaluating workgroups of the last launch (or, in a persistent launch, of its last pass)
// stamped - [grid][8] phases of wave 0, [grid][16] tile-loop ends per wave, [grid][8] prologue points (NPBNN_EXP_PROLOGUE_STAMPS builds).
// first_wg: workgroups before it do not evaluate (the step workgroup of the flag-ordered schedules).  Frees the buffer.
void report_eval_stamps(unsigned long long* d_stamps, int grid, int wpb, int first_wg) {
    std::vector<unsigned long long> hs((size_t)grid * 32);
    (void)hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_stamps);
    const int nwg = grid - first_wg;
    if (nwg < 1) return;
    if (getenv("NPBNN_EVAL_STAMPS") && atoi(getenv("NPBNN_EVAL_STAMPS")) >= 2) {
        // per workgroup: pass start -> end of the pass (prologue stamp 4), the tile phase of wave 0, by (b - first_wg) % 8 (its place in the
        // round-robin over the 8 XCDs)
        auto start_of_wg = [&](int b) { const unsigned long long* q = &hs[(size_t)b * 8]; return q[0] <= q[1] ? q[0] : q[1]; };
        double by_xcd[8] = {0}, tiles_xcd[8] = {0};
        int n_xcd[8] = {0};
        std::vector<std::pair<double, int>> dur;
        for (int b = first_wg; b < grid; ++b) {
            const double d = (double)(hs[(size_t)b * 8 + 6] - start_of_wg(b)) * 0.01;      // (start: see below)
            const double t = (double)(hs[(size_t)b * 8 + 4] - hs[(size_t)b * 8 + 3]) * 0.01;
            dur.push_back({d, b});
            by_xcd[b % 8] += d; tiles_xcd[b % 8] += t; ++n_xcd[b % 8];
        }
        std::sort(dur.begin(), dur.end());
        fprintf(stderr, "[npbnn eval stamps] pass start -> end per workgroup, us: fastest %.2f (wg %d), median %.2f, slowest", dur.front().first, dur.front().second,
                dur[dur.size() / 2].first);
        for (size_t i = dur.size() >= 6 ? dur.size() - 6 : 0; i < dur.size(); ++i) fprintf(stderr, " %.2f (wg %d)", dur[i].first, dur[i].second);
        fprintf(stderr, "\n[npbnn eval stamps] mean by blockIdx %% 8:");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.2f/%.2f", n_xcd[x] ? by_xcd[x] / n_xcd[x] : 0.0, n_xcd[x] ? tiles_xcd[x] / n_xcd[x] : 0.0);
        fprintf(stderr, " (pass/tile phase)\n");
        const int b = first_wg;
        fprintf(stderr, "[npbnn eval stamps] raw, workgroup %d:", b);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %llu", hs[(size_t)b * 8 + k]);
        fprintf(stderr, " | prologue:");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %llu", hs[(size_t)grid * 24 + (size_t)b * 8 + k]);
        fprintf(stderr, "\n");
    }
    {   // when each wave of a workgroup finished its tiles, relative to the workgroup's tile-loop start (mean over workgroups)
        double done[16] = {0};
        for (int b = first_wg; b < grid; ++b)
            for (int w = 0; w < wpb && w < 16; ++w)
                done[w] += (double)(hs[(size_t)grid * 8 + (size_t)b * 16 + w] - hs[(size_t)b * 8 + 3]) * 0.01;
        fprintf(stderr, "[npbnn eval stamps] tiles done per wave, us after the tile loop starts:");
        for (int w = 0; w < wpb && w < 16; ++w) fprintf(stderr, " %.1f", done[w] / nwg);
        fprintf(stderr, "\n");
    }
    // (a pass in the middle of a persistent launch: stamp 0 is then the start of the turn AFTER it - the launch's last, empty one - and
    // the pass is taken to start where its first barrier is reached, stamp 1; "issue" is not known for it)
    auto start_of = [&](const unsigned long long* q) { return q[0] <= q[1] ? q[0] : q[1]; };
    unsigned long long first = ~0ull, last = 0, first_end = ~0ull;
    double acc[8] = {0};
    for (int b = first_wg; b < grid; ++b) {
        const unsigned long long* q = &hs[(size_t)b * 8];
        if (start_of(q) < first) first = start_of(q);
        if (q[6] > last) last = q[6];
        if (q[6] < first_end) first_end = q[6];
        if (q[0] <= q[1]) acc[1] += (double)(q[1] - q[0]) * 0.01;                     // 100 MHz wall clock -> us
        for (int k = 2; k <= 6; ++k) acc[k] += (double)(q[k] - q[k - 1]) * 0.01;
        acc[7] += (double)q[7] * 0.01;
    }
    if (hs[(size_t)grid * 24 + (size_t)first_wg * 8] || hs[(size_t)grid * 24 + (size_t)first_wg * 8 + 3]) {      // (a build with NPBNN_EXP_PROLOGUE_STAMPS)
        double px[5] = {0};
        for (int b = first_wg; b < grid; ++b)
            for (int k = 0; k < 5; ++k) px[k] += (double)(hs[(size_t)grid * 24 + (size_t)b * 8 + k] - hs[(size_t)b * 8]) * 0.01;
        double period = 0, pmax = 0;
        int np_ = 0;
        for (int b = first_wg; b < grid; ++b) {
            const unsigned long long prev = hs[(size_t)grid * 24 + (size_t)b * 8 + 5];
            if (prev) { const double d = (double)(hs[(size_t)b * 8] - prev) * 0.01; period += d; if (d > pmax) pmax = d; ++np_; }
        }
        if (np_) fprintf(stderr, "[npbnn eval stamps] persistent launch: start of the pass before -> start of this one, us: mean %.2f, slowest workgroup %.2f\n", period / np_, pmax);
        fprintf(stderr, "[npbnn eval stamps] prologue of wave 0, us after its start: parameters read %.2f, pass descriptor %.2f, image copies requested %.2f, "
                        "first X pieces requested %.2f; end of the pass (sums out, workgroup reported done) %.2f\n", px[0] / nwg, px[1] / nwg, px[2] / nwg, px[3] / nwg, px[4] / nwg);
    }
    double late = 0;
    for (int b = first_wg; b < grid; ++b) late += (double)(start_of(&hs[(size_t)b * 8]) - first) * 0.01;
    fprintf(stderr, "[npbnn eval stamps] wave 0 of a workgroup, mean us: start skew %.2f | issue %.2f  barrier1 %.2f  patch %.2f  tiles %.2f  "
                    "barrier2 %.2f  partials %.2f | tails within tiles %.2f | first start -> first end %.2f, -> last end %.2f\n",
            late / nwg, acc[1] / nwg, acc[2] / nwg, acc[3] / nwg, acc[4] / nwg, acc[5] / nwg, acc[6] / nwg, acc[7] / nwg,
            (double)(first_end - first) * 0.01, (double)(last - first) * 0.01);
}

EvalParams make_params(npbnn_ctx* ctx, const Dataset& d) {
    EvalParams p{};
    p.X = ctx->net.l0_f16 ? d.X16 : d.X;
    p.labels = d.labels;
    p.targets = d.targets;
    p.inst_w = nullptr;
    p.image = ctx->d_image;
    p.n_rows = d.n_rows;
    p.n_tiles = d.n_tiles;
    p.has_pass = 0;
    p.Fp = ctx->net.l0_f16 ? d.Fp16 : d.Fp;
    p.net = ctx->net;
    p.lay = layout_for(ctx, d);
    p.cand_slopes = nullptr;
    return p;
}

int check_dataset_for_lik(npbnn_ctx* ctx, const Dataset& d, int lik) {
    if (!d.X) return fail(ctx, NPBNN_E_STATE, "no data matrix for this set");
    if (d.F != ctx->arch.in_dim)
        return fail(ctx, NPBNN_E_ARG, "data has %d features but the network expects %d", d.F, ctx->arch.in_dim);
    if (lik == NPBNN_LIK_CATEGORICAL && !d.labels) return fail(ctx, NPBNN_E_STATE, "categorical likelihood needs labels (npbnn_set_labels_i64)");
    if (lik == NPBNN_LIK_GAUSS || lik_needs_row_scratch(lik)) {
        if (!d.targets) return fail(ctx, NPBNN_E_STATE, "this likelihood needs targets (npbnn_set_targets_f64)");
        if (d.k != ctx->net.k_targets) return fail(ctx, NPBNN_E_ARG, "targets have %d columns, architecture says %d", d.k, ctx->net.k_targets);
    }
    return NPBNN_OK;
}

int rebuild_net(npbnn_ctx* ctx, bool f16) {
    int rc = build_net(ctx, &ctx->arch, f16);
    if (rc) return rc;
    if (ctx->d_image) { (void)hipFree(ctx->d_image); ctx->d_image = nullptr; }
    if (ctx->d_w2img) { (void)hipFree(ctx->d_w2img); ctx->d_w2img = nullptr; }
    if (ctx->d_w2scale) { (void)hipFree(ctx->d_w2scale); ctx->d_w2scale = nullptr; }
    if (ctx->wide) return wide_build(ctx, f16);
    wide_free(ctx);
    size_t lds = 0;
    if (pick_waves_per_block(ctx, &lds, 1, make_wave_layout(true, false, ctx->net.k_targets, ctx->net.L[0].kt, ctx->net.lik_kind)) == 0)
        return fail(ctx, NPBNN_E_ARG, "network too large: weight image of %d KiB does not fit the %zu KiB LDS of a CU",
                    ctx->net.image_floats * 4 / 1024, ctx->lds_limit / 1024);
    // (room for kMaxCand independent images: npbnn_predict_sets stages that many weight sets per pass)
    HIP_TRY(ctx, hipMalloc(&ctx->d_image, (size_t)kMaxCand * ctx->net.image_floats * sizeof(float)));
    HIP_TRY(ctx, hipMemset(ctx->d_image, 0, (size_t)kMaxCand * ctx->net.image_floats * sizeof(float)));
    // where each packed weight lives in the image (bias column -> bias slot, else its MFMA fragment slot)
    std::vector<int> map((size_t)ctx->n_weights);
    std::vector<float> scale;
    std::vector<float> wscale;
    if (f16) {
        scale.assign((size_t)ctx->n_weights, 1.0f);
        wscale.resize((size_t)round_up(ctx->arch.in_dim, 32));
        HIP_TRY(ctx, hipMemcpy(wscale.data(), ctx->d_wscale, wscale.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (int l = 0; l < ctx->net.n_layers; ++l) {
        const LayerMeta& L = ctx->net.L[l];
        const int ld = L.in_dim + L.has_bias;
        for (int o = 0; o < L.out_dim; ++o)
            for (int j = 0; j < ld; ++j) {
                const size_t wi = (size_t)L.w_off + (size_t)o * ld + j;
                int pos;
                const bool in_perm = l >= 1 && ctx->net.L[l - 1].out_perm;      // (a permuted layer has a single tile: o, c < 16)
                const int rows = f16 ? ctx->net.l0_rows : 16;         // (NetMeta::l0_rows: layer 0's units per tile in this image)
                if (L.has_bias && j == 0) pos = L.bias_off + (L.out_perm ? tile_pos(o) : (l == 0 && rows < 16) ? l0_pos_of_unit(o, rows) : o);
                else {
                    int c = in_perm ? tile_pos(j - L.has_bias) : j - L.has_bias;
                    if (l == 1 && rows < 16) c = l0_pos_of_unit(c, rows);       // (the position layer 0 leaves that unit at)
                    int mt = o / 16, u = L.out_perm ? tile_pos(o) : o % 16;
                    if (l == 0 && f16) {
                        mt = o / rows; u = o % rows;
                        const int ks = c / 32, kg = (c % 32) / 8, jj = c % 8;
                        if (ks < ctx->net.l0_begin[mt] || ks >= ctx->net.l0_end[mt]) pos = kSkipPos;     // outside the block structure: always 0
                        else {
                            const int slot = ctx->net.l0_base[mt] + ks - ctx->net.l0_begin[mt];
                            const int half_index = 2 * L.frag_off + ((slot * 2) * (4 * rows) + kg * rows + u) * 8 + jj;
                            pos = (int)(0x80000000u | (rows < 16 ? (unsigned)kPosCompact : 0u) | (unsigned)half_index);
                        }
                        scale[wi] = wscale[(size_t)c];
                    } else if (l == 0) {
                        const int kt = c / 16, kq = (c % 16) / 4, sidx = c % 4;
                        if (kt < ctx->net.l0_begin[mt] || kt >= ctx->net.l0_end[mt]) pos = kSkipPos;
                        else pos = L.frag_off + ((ctx->net.l0_base[mt] + kt - ctx->net.l0_begin[mt]) * 64 + kq * 16 + u) * 4 + sidx;
                    } else if (l == 1 && ctx->net.l1_f16) {      // (layer 0 is not permuted: c is the unit; scale stays 1)
                        const int t_in = c / 16, kq = (c % 16) / 4, q = t_in / 2, e = 4 * (t_in & 1) + c % 4;
                        const int half_index = 2 * L.frag_off + ((q * 2) * 64 + kq * 16 + u) * 8 + e;
                        pos = (int)(0x80000000u | (unsigned)half_index);
                    } else {
                        const int kt = c / 16, kq = (c % 16) / 4, sidx = c % 4;
                        pos = L.frag_off + ((kt * L.mt + mt) * 64 + kq * 16 + u) * 4 + sidx;
                    }
                }
                map[wi] = pos;
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

double wall_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void launch_pack_weights(npbnn_ctx* ctx, const double* d_w, const double* d_col_override, float* image, int* flags) {
    if (ctx->wide) { wide_pack(ctx, d_w, d_col_override, image, flags); return; }
    const int total = pack_item_count(ctx->net, true);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, d_w, d_col_override,
                       ctx->n_classw ? ctx->d_classw : nullptr, image, ctx->net, ctx->net.l0_f16 ? ctx->d_wscale : nullptr, flags);
}

int launch_plain_eval(npbnn_ctx* ctx, const LaunchPlan& lp, int which) {
    if (lp.wide) return wide_forward(ctx, which, ctx->d_image, false);
    hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
    return NPBNN_OK;
}

// confusion counts: device and pinned host buffers for n_classes x n_classes
int ensure_conf(npbnn_ctx* ctx, int n_classes) {
    unsigned need = (unsigned)(n_classes < kResidentMaxWidth ? kResidentMaxWidth : n_classes);
    if (need <= ctx->conf_cap) return NPBNN_OK;
    if (ctx->d_conf) { (void)hipFree(ctx->d_conf); ctx->d_conf = nullptr; }
    if (ctx->h_conf) { (void)hipHostFree(ctx->h_conf); ctx->h_conf = nullptr; }
    ctx->conf_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_conf, (size_t)need * need * sizeof(unsigned)));
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_conf, (size_t)need * need * sizeof(unsigned)));
    ctx->conf_cap = need;
    return NPBNN_OK;
}

void launch_finalize(npbnn_ctx* ctx) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, (const FinalizeParams*)ctx->d_fparams);
}

}  // namespace npbnn_api

extern "C" void npbnn_set_global_error_(const char* msg) { g_last_error = msg ? msg : ""; }

namespace npbnn_api {
void destroy_ctx(npbnn_ctx* c) {
    (void)hipSetDevice(c->device);
    wide_free(c);
    free_dataset(c->ds[0]);
    free_dataset(c->ds[1]);
    if (c->d_classw) (void)hipFree(c->d_classw);
    if (c->d_wraw) (void)hipFree(c->d_wraw);
    if (c->d_colov) (void)hipFree(c->d_colov);
    if (c->d_xscale) (void)hipFree(c->d_xscale);
    if (c->d_wscale) (void)hipFree(c->d_wscale);
    if (c->d_overflow) (void)hipFree(c->d_overflow);
    if (c->d_eparams) (void)hipFree(c->d_eparams);       // (d_fparams / d_cparams live in the same allocation)
    if (c->d_xbuf) (void)hipFree(c->d_xbuf);
    if (c->h_xbuf) (void)hipHostFree(c->h_xbuf);
    if (c->ev_x) (void)hipEventDestroy(c->ev_x);
    for (int i = 0; i < 2; ++i) {
        if (c->stream_e[i]) (void)hipStreamDestroy(c->stream_e[i]);
    }
    if (c->h_params) (void)hipHostFree(c->h_params);
    if (c->d_gparams) (void)hipFree(c->d_gparams);
    if (c->h_gparams) (void)hipHostFree(c->h_gparams);
    if (c->d_w2scale) (void)hipFree(c->d_w2scale);
    if (c->d_image) (void)hipFree(c->d_image);
    if (c->d_w2img) (void)hipFree(c->d_w2img);
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_conf) (void)hipFree(c->d_conf);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_y) (void)hipFree(c->d_y);
    if (c->h_w) (void)hipHostFree(c->h_w);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_conf) (void)hipHostFree(c->h_conf);
    void* chain_bufs[] = {c->d_spec, c->d_spec_pv, c->d_spec_touch, c->d_spec_part, c->d_res, c->d_pv, c->d_mask, c->d_idx, c->d_pos, c->d_pscale, c->d_smult, c->d_hast, c->d_pscale_w, c->d_slopes, c->d_sidx, c->d_sdelta, c->d_shard_recv, c->d_shard_part};
    for (void* b : chain_bufs)
        if (b) (void)hipFree(b);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_shard) (void)hipHostFree(c->h_shard);
    if (c->ev[0]) (void)hipEventDestroy(c->ev[0]);
    if (c->ev[1]) (void)hipEventDestroy(c->ev[1]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
}  // namespace npbnn_api

extern "C" {

int npbnn_abi_version(void) { return NPBNN_ABI_VERSION; }

int npbnn_device_count(int* out) {
    if (!out) return fail(nullptr, NPBNN_E_ARG, "null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        return fail(nullptr, NPBNN_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out = n;
    return NPBNN_OK;
}

const char* npbnn_last_error(const npbnn_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int npbnn_create(int device_id, npbnn_ctx** out) {
    if (!out) return fail(nullptr, NPBNN_E_ARG, "null out");
    *out = nullptr;
    int n = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(nullptr, NPBNN_E_ARG, "device %d not present (%d devices)", device_id, n);
    HIP_TRY(nullptr, hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, NPBNN_E_ARG, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    npbnn_ctx* c = new npbnn_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    c->lds_limit = 160 * 1024;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&c->d_conf, (size_t)kResidentMaxWidth * kResidentMaxWidth * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&c->d_out, sizeof(npbnn_eval_out));
    if (e == hipSuccess) e = hipMalloc(&c->d_overflow, sizeof(int));
    // the kernels' parameter blocks: ONE device allocation laid out like its page-locked staging twin, so that a chain batch sends
    // all of them in one copy
    if (e == hipSuccess) e = hipMalloc(&c->d_eparams, sizeof(EvalParams) + sizeof(FinalizeParams) + sizeof(ChainParams));
    if (e == hipSuccess) {
        c->d_fparams = reinterpret_cast<FinalizeParams*>(reinterpret_cast<char*>(c->d_eparams) + sizeof(EvalParams));
        c->d_cparams = reinterpret_cast<ChainParams*>(reinterpret_cast<char*>(c->d_eparams) + sizeof(EvalParams) + sizeof(FinalizeParams));
    }
    if (e == hipSuccess) e = hipHostMalloc(&c->h_params, sizeof(EvalParams) + sizeof(FinalizeParams) + sizeof(ChainParams));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_out, sizeof(npbnn_eval_out));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_conf, (size_t)kResidentMaxWidth * kResidentMaxWidth * sizeof(unsigned));
    if (e == hipSuccess) c->conf_cap = kResidentMaxWidth;
    if (e == hipSuccess) e = hipEventCreate(&c->ev[0]);
    if (e == hipSuccess) e = hipEventCreate(&c->ev[1]);
    if (e != hipSuccess) {
        npbnn_destroy(c);
        return fail(nullptr, NPBNN_E_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return NPBNN_OK;
}

void npbnn_destroy(npbnn_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->data_owner) unshare_data(c);
    if (c->n_borrowers > 0) {        // others still read this context's matrices: it goes when the last of them does
        c->zombie = true;
        return;
    }
    destroy_ctx(c);
}

int npbnn_share_data(npbnn_ctx* ctx, npbnn_ctx* owner) {
    if (!ctx || !owner || ctx == owner) return fail(ctx, NPBNN_E_ARG, "share_data: bad arguments");
    while (owner->data_owner) owner = owner->data_owner;          // the root holds the memory
    if (owner == ctx) return fail(ctx, NPBNN_E_ARG, "share_data: contexts borrow from each other");
    if (owner->device != ctx->device) return fail(ctx, NPBNN_E_ARG, "share_data: contexts on devices %d and %d", ctx->device, owner->device);
    if (ctx->n_borrowers > 0) return fail(ctx, NPBNN_E_STATE, "share_data: %d other context(s) use this one's matrices", ctx->n_borrowers);
    if (!owner->ds[0].X) return fail(ctx, NPBNN_E_STATE, "share_data: the owner has no training matrix");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the owner's fp16-split copies are built now (on its stream) so that borrowers never have to
    for (int w = 0; w < 2; ++w)
        if (owner->ds[w].X && owner->l0_option != NPBNN_L0_F32) {
            int usable = 0;
            int rc = ensure_x16(owner, w, &usable);
            if (rc) { ctx->err = owner->err; return rc; }
        }
    HIP_TRY(ctx, hipStreamSynchronize(owner->stream));
    if (ctx->data_owner) unshare_data(ctx);
    free_dataset(ctx->ds[0]);
    free_dataset(ctx->ds[1]);
    if (ctx->d_xscale) (void)hipFree(ctx->d_xscale);
    if (ctx->d_wscale) (void)hipFree(ctx->d_wscale);
    for (int w = 0; w < 2; ++w) {
        const Dataset& o = owner->ds[w];
        if (!o.X) continue;
        Dataset& d = ctx->ds[w];
        d.X = o.X; d.X16 = o.X16; d.n_rows = o.n_rows; d.n_tiles = o.n_tiles; d.F = o.F; d.Fp = o.Fp; d.Fp16 = o.Fp16;
        d.f16_state = o.f16_state;
        d.borrowed = true;
    }
    ctx->d_xscale = owner->d_xscale;
    ctx->f16_shifted_cols = owner->f16_shifted_cols;
    ctx->f16_max_shift = owner->f16_max_shift;
    ctx->d_wscale = owner->d_wscale;
    ctx->scale_F = owner->scale_F;
    ctx->data_owner = owner;
    owner->n_borrowers += 1;
    ctx->arch_set = false;            // (layer-0 layout depends on the data: set_arch again)
    return NPBNN_OK;
}

int npbnn_set_data_f64(npbnn_ctx* ctx, const double* X, int64_t n_rows, int32_t F, int which) {
    return upload_matrix<double>(ctx, X, n_rows, F, which);
}

int npbnn_set_data_f32(npbnn_ctx* ctx, const float* X, int64_t n_rows, int32_t F, int which) {
    return upload_matrix<float>(ctx, X, n_rows, F, which);
}

int npbnn_set_labels_i64(npbnn_ctx* ctx, const int64_t* y, int64_t n_rows, int which) {
    int rc = check_rows(ctx, which, n_rows, "set_labels");
    if (rc) return rc;
    if (!y) return fail(ctx, NPBNN_E_ARG, "set_labels: null labels");
    Dataset& d = ctx->ds[which];
    const size_t n_pad = (size_t)d.n_tiles * 16;
    std::vector<int> tmp(n_pad, -1);
    for (int64_t i = 0; i < n_rows; ++i) {
        if (y[i] < 0 || y[i] >= NPBNN_MAX_WIDTH)
            return fail(ctx, NPBNN_E_ARG, "set_labels: label %lld at row %lld outside 0..%d", (long long)y[i], (long long)i, NPBNN_MAX_WIDTH - 1);
        tmp[i] = (int)y[i];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!d.labels) HIP_TRY(ctx, hipMalloc(&d.labels, n_pad * sizeof(int)));
    HIP_TRY(ctx, hipMemcpy(d.labels, tmp.data(), n_pad * sizeof(int), hipMemcpyHostToDevice));
    return NPBNN_OK;
}

int npbnn_set_targets_f64(npbnn_ctx* ctx, const double* Y, int64_t n_rows, int32_t k, int which) {
    int rc = check_rows(ctx, which, n_rows, "set_targets");
    if (rc) return rc;
    if (!Y || k < 1 || k > NPBNN_MAX_TARGETS) return fail(ctx, NPBNN_E_ARG, "set_targets: need 1..%d target columns, got %d", NPBNN_MAX_TARGETS, k);
    Dataset& d = ctx->ds[which];
    const size_t n_pad = (size_t)d.n_tiles * 16;
    std::vector<float> tmp(n_pad * k, 0.0f);
    for (size_t i = 0; i < (size_t)n_rows * k; ++i) tmp[i] = (float)Y[i];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (d.targets && d.k != k) { (void)hipFree(d.targets); d.targets = nullptr; }
    if (!d.targets) HIP_TRY(ctx, hipMalloc(&d.targets, n_pad * k * sizeof(float)));
    d.k = k;
    HIP_TRY(ctx, hipMemcpy(d.targets, tmp.data(), n_pad * k * sizeof(float), hipMemcpyHostToDevice));
    return NPBNN_OK;
}

int npbnn_set_row_weights(npbnn_ctx* ctx, const double* instance_w, int64_t n_rows, const double* class_w, int32_t n_classes) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    Dataset& d = ctx->ds[0];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (instance_w) {
        int rc = check_rows(ctx, 0, n_rows, "set_row_weights");
        if (rc) return rc;
        const size_t n_pad = (size_t)d.n_tiles * 16;
        std::vector<float> tmp(n_pad, 0.0f);
        for (int64_t i = 0; i < n_rows; ++i) tmp[i] = (float)instance_w[i];
        if (!d.inst_w) HIP_TRY(ctx, hipMalloc(&d.inst_w, n_pad * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(d.inst_w, tmp.data(), n_pad * sizeof(float), hipMemcpyHostToDevice));
    } else if (d.inst_w) {
        (void)hipFree(d.inst_w);
        d.inst_w = nullptr;
    }
    if (class_w) {
        if (n_classes < 1 || n_classes > NPBNN_MAX_WIDTH) return fail(ctx, NPBNN_E_ARG, "set_row_weights: n_classes=%d", n_classes);
        if (!ctx->d_classw) HIP_TRY(ctx, hipMalloc(&ctx->d_classw, NPBNN_MAX_WIDTH * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_classw, class_w, (size_t)n_classes * sizeof(double), hipMemcpyHostToDevice));
        ctx->n_classw = n_classes;
    } else {
        ctx->n_classw = 0;
    }
    if (ctx->arch_set && (ctx->n_classw > 0) != (ctx->net.classw_off >= 0))      // the image gains / loses its class-weight block
        return rebuild_net(ctx, ctx->net.l0_f16 != 0);
    return NPBNN_OK;
}

int npbnn_set_arch(npbnn_ctx* ctx, const npbnn_arch* arch) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!arch) return fail(ctx, NPBNN_E_ARG, "null arch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->arch_set = false;
    const npbnn_arch previous = ctx->arch;
    ctx->arch = *arch;
    ctx->l0_blocks.clear();                      // (a structure belongs to one architecture: npbnn_set_layer_mask after this call)
    int rc = rebuild_net(ctx, false);            // float32 layout first; plan_launch switches to fp16-split when it applies
    if (rc) {
        ctx->arch = previous;
        return rc;
    }
    if (ctx->d_wraw) { (void)hipFree(ctx->d_wraw); ctx->d_wraw = nullptr; }
    if (ctx->d_colov) { (void)hipFree(ctx->d_colov); ctx->d_colov = nullptr; }
    if (ctx->h_w) { (void)hipHostFree(ctx->h_w); ctx->h_w = nullptr; }
    if (ctx->d_wcur) { (void)hipFree(ctx->d_wcur); ctx->d_wcur = nullptr; }
    if (ctx->d_mask) { (void)hipFree(ctx->d_mask); ctx->d_mask = nullptr; }
    if (ctx->d_pscale_w) { (void)hipFree(ctx->d_pscale_w); ctx->d_pscale_w = nullptr; }
    HIP_TRY(ctx, hipMalloc(&ctx->d_wraw, (size_t)kMaxCand * ctx->n_weights * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_colov, (size_t)arch->in_dim * sizeof(double)));
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_w, ((size_t)ctx->n_weights + arch->in_dim) * sizeof(double)));
    ctx->arch_set = true;
    return NPBNN_OK;
}

int npbnn_set_layer_mask(npbnn_ctx* ctx, const double* mask_packed) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "set_layer_mask: call npbnn_set_arch first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned char> blocks;
    if (mask_packed) {
        const npbnn_arch& a = ctx->arch;
        const int in = a.in_dim, out = a.out_dim[0], ld = in + (a.has_bias[0] ? 1 : 0);
        const int g16 = (in + 15) / 16, mts = (out + 15) / 16;
        blocks.assign((size_t)mts * g16, 0);
        bool dense = true;
        for (int o = 0; o < out; ++o)
            for (int c = 0; c < in; ++c)
                if (mask_packed[(size_t)o * ld + (a.has_bias[0] ? 1 : 0) + c] != 0.0) blocks[(size_t)(o / 16) * g16 + c / 16] = 1;
        for (unsigned char b : blocks) dense = dense && b != 0;
        if (dense) blocks.clear();
    }
    if (blocks == ctx->l0_blocks) return NPBNN_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->l0_blocks.swap(blocks);
    return rebuild_net(ctx, ctx->net.l0_f16 != 0);
}

int npbnn_set_option(npbnn_ctx* ctx, int option, int value) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (option == NPBNN_OPT_L0_PRECISION) {
        if (value < NPBNN_L0_AUTO || value > NPBNN_L0_F16) return fail(ctx, NPBNN_E_ARG, "set_option: layer-0 precision %d", value);
        ctx->l0_option = value;
        return NPBNN_OK;
    }
    if (option == NPBNN_OPT_FAST_TAILS) {
        ctx->fast_option = value ? 1 : 0;
        return NPBNN_OK;
    }
    if (option == NPBNN_OPT_PERSISTENT) {
        ctx->persist_option = value ? 1 : 0;
        return NPBNN_OK;
    }
    if (option == NPBNN_OPT_WIDE) {
        const int on = value ? 1 : 0;
        if (on != ctx->wide_option) {
            ctx->wide_option = on;
            if (ctx->arch_set) {
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                return rebuild_net(ctx, ctx->net.l0_f16 != 0);
            }
        }
        return NPBNN_OK;
    }
    if (option == NPBNN_OPT_TRAINABLE_SLOPES) {
        const int on = value ? 1 : 0;
        if (on != ctx->slopes_option) {
            ctx->slopes_option = on;
            if (ctx->arch_set) return rebuild_net(ctx, ctx->net.l0_f16 != 0);      // (the image layout changes)
        }
        return NPBNN_OK;
    }
    return fail(ctx, NPBNN_E_ARG, "set_option: unknown option %d", option);
}

int npbnn_get_info(npbnn_ctx* ctx, int what, int* out) {
    if (!ctx || !out) return fail(ctx, NPBNN_E_ARG, "get_info: bad arguments");
    if (what == NPBNN_INFO_L0_F16) { *out = ctx->net.l0_f16; return NPBNN_OK; }
    if (what == NPBNN_INFO_WIDE) { *out = (ctx->arch_set && ctx->wide) ? 1 : 0; return NPBNN_OK; }
    if (what == NPBNN_INFO_F16_MOVED_COLUMNS) { *out = ctx->f16_shifted_cols; return NPBNN_OK; }
    if (what == NPBNN_INFO_F16_MAX_MOVE) { *out = ctx->f16_max_shift; return NPBNN_OK; }
    if (ctx->arch_set && ctx->wide && (what == NPBNN_INFO_WAVES_PER_BLOCK || what == NPBNN_INFO_MAX_CANDIDATES || what == NPBNN_INFO_FAST_TAILS)) {
        *out = what == NPBNN_INFO_WAVES_PER_BLOCK ? 4 : what == NPBNN_INFO_MAX_CANDIDATES ? 1 : 0;      // (group passes and prediction sets: one weight set per pass)
        return NPBNN_OK;
    }
    if (what == NPBNN_INFO_WAVES_PER_BLOCK) { size_t lds = 0; *out = pick_waves_per_block(ctx, &lds, 1, layout_for(ctx, ctx->ds[0])); return NPBNN_OK; }
    if (what == NPBNN_INFO_N_CU) { *out = ctx->n_cu; return NPBNN_OK; }
    if (what == NPBNN_INFO_TURN_NS_OVERLAPPED || what == NPBNN_INFO_TURN_NS_BETWEEN) {
        *out = (int)(1000.0 * ctx->turn_us[what == NPBNN_INFO_TURN_NS_BETWEEN ? 1 : 0]);
        return NPBNN_OK;
    }
    if (what == NPBNN_INFO_IT_NS_OVERLAPPED || what == NPBNN_INFO_IT_NS_BETWEEN) {
        const double* c = ctx->it_us[what == NPBNN_INFO_IT_NS_BETWEEN ? 1 : 0];      // (short batches - dispatches of 100 - when measured, else long ones)
        *out = (int)(1000.0 * (c[0] > 0.0 ? c[0] : c[1]));
        return NPBNN_OK;
    }
    if (what == NPBNN_INFO_MAX_CANDIDATES) {        // what plan_launch would give a chain pass that asks for as many as fit
        *out = 1;
        if (!ctx->arch_set || !ctx->ds[0].X) return NPBNN_OK;
        int n = (max_inner_tiles(ctx->net) == 1 && !lik_needs_row_scratch(ctx->net.lik_kind)) ? kMaxCand : 1;
        if (n > max_cand_for(ctx->net.L[0].mt)) n = max_cand_for(ctx->net.L[0].mt);
        const WaveLayout lay = layout_for(ctx, ctx->ds[0], false);
        const bool fast = fast_launch_ok(ctx, ctx->ds[0]);
        size_t lds = 0;
        while (n > 1 && pick_waves_per_block(ctx, &lds, n, lay, false, fast) < 8) --n;
        *out = n;
        return NPBNN_OK;
    }
    if (what == NPBNN_INFO_FAST_TAILS) { *out = (ctx->arch_set && fast_launch_ok(ctx, ctx->ds[0])) ? 1 : 0; return NPBNN_OK; }
    return fail(ctx, NPBNN_E_ARG, "get_info: unknown item %d", what);
}

static int eval_once(npbnn_ctx* ctx, const double* W_packed, const double* act_prm, const double* col_override, double lik_temp,
                     const double* sigma, int which, npbnn_eval_out* out, int64_t* confusion, int force_f32, int* overflowed) {
    const int lik = ctx->net.lik_kind;
    Dataset& d = ctx->ds[which];
    LaunchPlan lp;
    int rc = plan_launch(ctx, which, &lp, force_f32, 1, false, confusion == nullptr, true);
    if (rc) return rc;
    rc = ensure_work_buffers(ctx, lp.n_waves);
    if (rc) return rc;
    rc = stage_weights(ctx, W_packed, act_prm, col_override);
    if (rc) return rc;
    EvalParams p = make_params(ctx, d);
    p.partials = ctx->d_partials;
    p.inst_w = (which == 0) ? d.inst_w : nullptr;
    p.use_classw = (which == 0 && ctx->n_classw > 0) ? 1 : 0;
    const int C = ctx->net.n_out;
    if (confusion) {
        rc = ensure_conf(ctx, C);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_conf, 0, (size_t)C * C * sizeof(unsigned), ctx->stream));
        p.confusion = ctx->d_conf;
    }
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    rc = launch_plain_eval(ctx, lp, which);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    FinalizeParams f{};
    f.partials = ctx->d_partials;
    f.n_waves = lp.n_waves;
    f.lik_kind = lik;
    f.k_targets = ctx->net.k_targets;
    f.n_rows = d.n_rows;
    f.lik_temp = lik_temp;
    f.sigma_given = sigma ? 1 : 0;
    if (sigma)
        for (int j = 0; j < ctx->net.k_targets; ++j) f.sigma[j] = sigma[j];
    f.out = ctx->d_out;
    rc = push_finalize_params(ctx, f);
    if (rc) return rc;
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, (const FinalizeParams*)ctx->d_fparams);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, sizeof(npbnn_eval_out), hipMemcpyDeviceToHost, ctx->stream));
    if (confusion)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_conf, ctx->d_conf, (size_t)C * C * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    int ovf = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ovf & kFlagStructure) return fail(ctx, NPBNN_E_ARG, "eval: a layer-0 weight is not zero where the mask given to npbnn_set_layer_mask is");
    *overflowed = ctx->net.l0_f16 && (ovf & kFlagF16Range);
    if (*overflowed) return NPBNN_OK;
    *out = *ctx->h_out;
    if (confusion)
        for (int i = 0; i < C * C; ++i) confusion[i] = (int64_t)ctx->h_conf[i];
    return NPBNN_OK;
}

int npbnn_eval(npbnn_ctx* ctx, const double* W_packed, const double* act_prm, const double* col_override, double lik_temp,
               const double* sigma, int which, npbnn_eval_out* out, int64_t* confusion) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!out) return fail(ctx, NPBNN_E_ARG, "eval: null out");
    if (which != 0 && which != 1) return fail(ctx, NPBNN_E_ARG, "eval: which must be 0 or 1");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "eval: call npbnn_set_arch first");
    const int lik = ctx->net.lik_kind;
    if (lik == NPBNN_LIK_NONE) return fail(ctx, NPBNN_E_STATE, "eval: architecture was set with NPBNN_LIK_NONE");
    int rc = check_dataset_for_lik(ctx, ctx->ds[which], lik);
    if (rc) return rc;
    if (confusion && lik != NPBNN_LIK_CATEGORICAL) return fail(ctx, NPBNN_E_ARG, "eval: confusion counts need the categorical likelihood");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int overflowed = 0;
    rc = eval_once(ctx, W_packed, act_prm, col_override, lik_temp, sigma, which, out, confusion, 0, &overflowed);
    if (rc) return rc;
    if (overflowed) {   // a scaled layer-0 weight left the fp16 range: this evaluation runs on the exact float32 path
        if (ctx->l0_option == NPBNN_L0_F16) return fail(ctx, NPBNN_E_RANGE, "eval: a layer-0 weight left the fp16 range");
        rc = eval_once(ctx, W_packed, act_prm, col_override, lik_temp, sigma, which, out, confusion, 1, &overflowed);
    }
    return rc;
}

int npbnn_predict(npbnn_ctx* ctx, const double* W_packed, const double* act_prm, const double* col_override, int which,
                  int apply_out_fn, double* out_y) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!out_y) return fail(ctx, NPBNN_E_ARG, "predict: null out_y");
    if (which != 0 && which != 1) return fail(ctx, NPBNN_E_ARG, "predict: which must be 0 or 1");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "predict: call npbnn_set_arch first");
    Dataset& d = ctx->ds[which];
    int rc = check_dataset_for_lik(ctx, d, NPBNN_LIK_NONE);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int C = ctx->net.n_out;
    const size_t n_el = (size_t)d.n_rows * C;
    std::vector<float> tmp(n_el);
    for (int attempt = 0; attempt < 2; ++attempt) {
    LaunchPlan lp;
    rc = plan_launch(ctx, which, &lp, attempt);
    if (rc) return rc;
    rc = stage_weights(ctx, W_packed, act_prm, col_override);
    if (rc) return rc;
    if (n_el > ctx->d_y_cap) {
        if (ctx->d_y) (void)hipFree(ctx->d_y);
        ctx->d_y = nullptr;
        ctx->d_y_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_y, n_el * sizeof(float)));
        ctx->d_y_cap = n_el;
    }
    EvalParams p = make_params(ctx, d);
    p.labels = nullptr;
    p.targets = nullptr;
    p.net.lik_kind = NPBNN_LIK_NONE;
    p.y_out = ctx->d_y;
    p.predict_mode = apply_out_fn ? 2 : 1;
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    rc = launch_plain_eval(ctx, lp, which);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ctx->d_y, n_el * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    int ovf = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ovf & kFlagStructure) return fail(ctx, NPBNN_E_ARG, "predict: a layer-0 weight is not zero where the mask given to npbnn_set_layer_mask is");
    if (!(ctx->net.l0_f16 && (ovf & kFlagF16Range))) break;
    if (ctx->l0_option == NPBNN_L0_F16) return fail(ctx, NPBNN_E_RANGE, "predict: a layer-0 weight left the fp16 range");
    }
    for (size_t i = 0; i < n_el; ++i) out_y[i] = (double)tmp[i];
    return NPBNN_OK;
}

int npbnn_predict_sets(npbnn_ctx* ctx, const double* W_sets, const double* act_prm_sets, int32_t n_sets, int which, int apply_out_fn,
                       double* out_y) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!W_sets || !out_y || n_sets < 1) return fail(ctx, NPBNN_E_ARG, "predict_sets: bad arguments");
    if (which != 0 && which != 1) return fail(ctx, NPBNN_E_ARG, "predict_sets: which must be 0 or 1");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "predict_sets: call npbnn_set_arch first");
    Dataset& d = ctx->ds[which];
    int rc = check_dataset_for_lik(ctx, d, NPBNN_LIK_NONE);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int C = ctx->net.n_out;
    const int n_act = ctx->net.n_layers - 1;
    const size_t per_set = (size_t)d.n_rows * C;
    const size_t wn = (size_t)ctx->n_weights;
    if (kMaxCand * per_set > ctx->d_y_cap) {
        if (ctx->d_y) (void)hipFree(ctx->d_y);
        ctx->d_y = nullptr;
        ctx->d_y_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_y, kMaxCand * per_set * sizeof(float)));
        ctx->d_y_cap = kMaxCand * per_set;
    }
    std::vector<float> tmp(kMaxCand * per_set);
    std::vector<double> wstage(kMaxCand * wn);
    int s0 = 0;
    while (s0 < n_sets) {
        // sets that share their activation slopes travel together, up to kMaxCand per streaming read of X
        int g = 1;
        while (s0 + g < n_sets && g < kMaxCand &&
               (!act_prm_sets || n_act == 0 ||
                memcmp(act_prm_sets + (size_t)(s0 + g) * n_act, act_prm_sets + (size_t)s0 * n_act, (size_t)n_act * sizeof(double)) == 0))
            ++g;
        for (int attempt = 0; attempt < 2; ++attempt) {
            LaunchPlan lp;
            rc = plan_launch(ctx, which, &lp, attempt, g, true);
            if (rc) return rc;
            if (lp.n_cand < g) g = lp.n_cand;          // (fewer images fit the LDS: the rest waits for the next round)
            memcpy(wstage.data(), W_sets + (size_t)s0 * wn, (size_t)g * wn * sizeof(double));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_wraw, wstage.data(), (size_t)g * wn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            for (int l = 0; l < kMaxLayers; ++l) ctx->net.act_prm[l] = 0.f;
            if (act_prm_sets)
                for (int l = 0; l < n_act; ++l) ctx->net.act_prm[l] = (float)act_prm_sets[(size_t)s0 * n_act + l];
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(int), ctx->stream));
            for (int j = 0; j < g; ++j)         // (the weight-streamed path carries one set per pass: g = 1)
                launch_pack_weights(ctx, ctx->d_wraw + (size_t)j * wn, nullptr, ctx->d_image + (size_t)j * ctx->net.image_floats, ctx->d_overflow);
            HIP_TRY(ctx, hipGetLastError());
            EvalParams p = make_params(ctx, d);
            p.labels = nullptr;
            p.targets = nullptr;
            p.net.lik_kind = NPBNN_LIK_NONE;
            p.y_out = ctx->d_y;
            p.predict_mode = apply_out_fn ? 2 : 1;
            p.weight_sets = 1;
            p.lay = layout_for(ctx, d, true);
            rc = push_eval_params(ctx, p);
            if (rc) return rc;
            rc = launch_plain_eval(ctx, lp, which);
            if (rc) return rc;
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ctx->d_y, (size_t)g * per_set * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
            int ovf = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&ovf, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ovf & kFlagStructure) return fail(ctx, NPBNN_E_ARG, "predict_sets: a layer-0 weight is not zero where the mask given to npbnn_set_layer_mask is");
            if (!(ctx->net.l0_f16 && (ovf & kFlagF16Range))) break;
            if (ctx->l0_option == NPBNN_L0_F16) return fail(ctx, NPBNN_E_RANGE, "predict_sets: a layer-0 weight left the fp16 range");
        }
        double* dst = out_y + (size_t)s0 * per_set;
        for (size_t i = 0; i < (size_t)g * per_set; ++i) dst[i] = (double)tmp[i];
        s0 += g;
    }
    return NPBNN_OK;
}

}  // extern "C"

extern "C" {

int npbnn_device_synchronize(int device_id) {
    HIP_TRY(nullptr, hipSetDevice(device_id));
    HIP_TRY(nullptr, hipDeviceSynchronize());
    return NPBNN_OK;
}

int npbnn_pinned_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(nullptr, NPBNN_E_ARG, "pinned_alloc: bad arguments");
    *out = nullptr;
    HIP_TRY(nullptr, hipHostMalloc(out, bytes));
    return NPBNN_OK;
}

void npbnn_pinned_free(void* ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int npbnn_time_pass(npbnn_ctx* ctx, const double* W_packed, int n_candidates, int iters, double* ms_kernel, int* used_candidates) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (iters < 1 || !ms_kernel || !W_packed) return fail(ctx, NPBNN_E_ARG, "time_pass: bad arguments");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "time_pass: call npbnn_set_arch first");
    const int lik = ctx->net.lik_kind;
    Dataset& d = ctx->ds[0];
    int rc = check_dataset_for_lik(ctx, d, lik);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchPlan lp;
    rc = plan_launch(ctx, 0, &lp, 0, n_candidates < 1 ? kMaxCand : n_candidates, false, true);
    if (rc) return rc;
    rc = ensure_work_buffers(ctx, lp.n_waves);
    if (rc) return rc;
    rc = stage_weights(ctx, W_packed, nullptr, nullptr);
    if (rc) return rc;
    PassDesc pd{};
    pd.t0 = 0;
    pd.n_cand = lp.n_cand;          // every candidate = the staged weights (empty patch lists): same work as a chain pass
    EvalParams p = make_params(ctx, d);
    p.partials = ctx->d_partials;
    p.inst_w = d.inst_w;
    p.use_classw = ctx->n_classw > 0 ? 1 : 0;
    p.has_pass = 1;
    p.pass_desc[0] = pd;
    p.pv = nullptr;
    p.pos = nullptr;
    p.pscale = nullptr;
    p.M = 0;
    unsigned long long* d_stamps = nullptr;
    if (getenv("NPBNN_EVAL_STAMPS")) {      // diagnostics: per-phase wall-clock stamps of the last launch
        HIP_TRY(ctx, hipMalloc(&d_stamps, (size_t)lp.grid * 32 * sizeof(unsigned long long)));     // [grid][8] wave 0 + [grid][16] per wave + [grid][8] prologue
        HIP_TRY(ctx, hipMemset(d_stamps, 0, (size_t)lp.grid * 32 * sizeof(unsigned long long)));
        p.stamps = d_stamps;
    }
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    if (lp.wide) {      // the pass of the weight-streamed path: its layers' products + the likelihood kernel, timed together
        const float* img = ctx->d_image;
        if (lp.n_cand > 1) {          // (every candidate = the staged weights, as on the resident path)
            rc = wide_cand_begin(ctx);
            if (rc) return rc;
            img = ctx->d_wide_cand;
        }
        for (int i = 0; i < 3 && !rc; ++i) rc = wide_forward(ctx, 0, img, false, false, nullptr, lp.n_cand);
        HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        for (int i = 0; i < iters && !rc; ++i) rc = wide_forward(ctx, 0, img, false, false, nullptr, lp.n_cand);
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (rc) return rc;
        float msw = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&msw, ctx->ev[0], ctx->ev[1]));
        *ms_kernel = (double)msw / iters;
        if (used_candidates) *used_candidates = lp.n_cand;
        return NPBNN_OK;
    }
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    *ms_kernel = (double)ms / iters;
    if (used_candidates) *used_candidates = lp.n_cand;
    if (getenv("NPBNN_TIME_PASS_ONE_BY_ONE")) {      // diagnostics: every launch between events of its own, the stream idle before it (what a profiler's trace shows)
        double sum = 0.0, mn = 1e30, mx = 0.0;
        for (int i = 0; i < iters; ++i) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
            hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
            HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            float one = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&one, ctx->ev[0], ctx->ev[1]));
            sum += one; if (one < mn) mn = one; if (one > mx) mx = one;
        }
        fprintf(stderr, "[npbnn time_pass] %d launches one by one: %.2f us mean (%.2f-%.2f); back to back %.2f us per launch\n", iters, 1e3 * sum / iters, 1e3 * mn,
                1e3 * mx, 1e3 * (double)ms / iters);
    }
    if (const char* ns = getenv("NPBNN_TIME_PASS_STREAMS")) {      // diagnostics: the same independent launches dealt over several streams
        const int n_streams = atoi(ns) > 1 ? (atoi(ns) > 4 ? 4 : atoi(ns)) : 1;
        hipStream_t ss[4];
        for (int i = 0; i < n_streams; ++i) HIP_TRY(ctx, hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = wall_us();
            for (int i = 0; i < iters; ++i)
                hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ss[i % n_streams], (const EvalParams*)ctx->d_eparams, 0, 1);
            for (int i = 0; i < n_streams; ++i) HIP_TRY(ctx, hipStreamSynchronize(ss[i]));
            if (rep) fprintf(stderr, "[npbnn time_pass] %d independent launches dealt over %d stream(s): %.2f us per launch (wall clock)\n", iters, n_streams,
                             (wall_us() - t0) / iters);
        }
        for (int i = 0; i < n_streams; ++i) (void)hipStreamDestroy(ss[i]);
    }
    if (d_stamps) report_eval_stamps(d_stamps, lp.grid, lp.wpb, 0);
    return NPBNN_OK;
}

int npbnn_time_wide(npbnn_ctx* ctx, const double* W_packed, int iters, double* ms_layer0, double* ms_pass, int* info) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (iters < 1 || !ms_layer0 || !ms_pass || !W_packed) return fail(ctx, NPBNN_E_ARG, "time_wide: bad arguments");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "time_wide: call npbnn_set_arch first");
    Dataset& d = ctx->ds[0];
    int rc = check_dataset_for_lik(ctx, d, ctx->net.lik_kind);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchPlan lp;
    rc = plan_launch(ctx, 0, &lp, 0, 1, false, true);
    if (rc) return rc;
    if (!lp.wide) return fail(ctx, NPBNN_E_STATE, "time_wide: this network runs on the LDS-resident path");
    rc = ensure_work_buffers(ctx, lp.n_waves);
    if (rc) return rc;
    rc = stage_weights(ctx, W_packed, nullptr, nullptr);
    if (rc) return rc;
    EvalParams p = make_params(ctx, d);
    p.partials = ctx->d_partials;
    p.inst_w = d.inst_w;
    p.use_classw = ctx->n_classw > 0 ? 1 : 0;
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    int geo[4] = {0, 0, 0, 0};
    for (int only0 = 1; only0 >= 0; --only0) {
        for (int i = 0; i < 3 && !rc; ++i) rc = wide_forward(ctx, 0, ctx->d_image, false, only0 != 0, geo);
        HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        for (int i = 0; i < iters && !rc; ++i) rc = wide_forward(ctx, 0, ctx->d_image, false, only0 != 0, nullptr);
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (rc) return rc;
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        *(only0 ? ms_layer0 : ms_pass) = (double)ms / iters;
    }
    if (info) for (int i = 0; i < 4; ++i) info[i] = geo[i];
    return NPBNN_OK;
}

int npbnn_time_eval(npbnn_ctx* ctx, const double* W_packed, int iters, double* ms_main_kernel, double* ms_total) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (iters < 1 || !ms_main_kernel || !ms_total) return fail(ctx, NPBNN_E_ARG, "time_eval: bad arguments");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "time_eval: call npbnn_set_arch first");
    const int lik = ctx->net.lik_kind;
    Dataset& d = ctx->ds[0];
    int rc = check_dataset_for_lik(ctx, d, lik);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchPlan lp;
    rc = plan_launch(ctx, 0, &lp, 0, 1, false, true, true);      // (the build npbnn_eval runs)
    if (rc) return rc;
    rc = ensure_work_buffers(ctx, lp.n_waves);
    if (rc) return rc;
    rc = stage_weights(ctx, W_packed, nullptr, nullptr);
    if (rc) return rc;
    EvalParams p = make_params(ctx, d);
    p.partials = ctx->d_partials;
    p.inst_w = d.inst_w;
    p.use_classw = ctx->n_classw > 0 ? 1 : 0;
    FinalizeParams f{};
    f.partials = ctx->d_partials;
    f.n_waves = lp.n_waves;
    f.lik_kind = lik;
    f.k_targets = ctx->net.k_targets;
    f.n_rows = d.n_rows;
    f.lik_temp = 1.0;
    f.out = ctx->d_out;
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    rc = push_finalize_params(ctx, f);
    if (rc) return rc;
    // (1) the dominant kernel alone: `iters` back-to-back launches between one pair of events (per-launch event pairs
    //     would add ~4 us of command-processor overhead to each 20 us kernel); includes the ~1.5 us launch boundary
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 3 && !rc; ++i) rc = launch_plain_eval(ctx, lp, 0);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int i = 0; i < iters && !rc; ++i) rc = launch_plain_eval(ctx, lp, 0);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (rc) return rc;
    float burst = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&burst, ctx->ev[0], ctx->ev[1]));
    const double sum = (double)burst;
    // (2) evaluation = eval kernel + finalize
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int i = 0; i < iters && !rc; ++i) {
        rc = launch_plain_eval(ctx, lp, 0);
        hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, (const FinalizeParams*)ctx->d_fparams);
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    float tot = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&tot, ctx->ev[0], ctx->ev[1]));
    *ms_main_kernel = sum / iters;
    *ms_total = (double)tot / iters;
    return NPBNN_OK;
}

}  // extern "C"
