// C ABI of the gfx950 backend (see include/npbnn_hip.h for the contract and the reference
// functions each entry point replaces).  Host-side orchestration only: device buffers, launches,
// the HIP stream of the chain.  No torch, no BLAS library, no CPU fallback for the numerics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "npbnn_hip.h"
#define NPBNN_KERNELS_MAIN
#include "npbnn_kernels.hip.h"

using namespace npbnn;

// the evaluation-kernel instantiations live in npbnn_eval_inst_*.hip (compiled in parallel)
namespace npbnn {
eval_fn_t pick_eval_mti8_cat(int mt0, int f16);
eval_fn_t pick_eval_mti8_gauss(int mt0, int f16);
eval_fn_t pick_eval_mti8_gen(int mt0, int f16);
eval_fn_t pick_eval_d1_cat(int mt0, int f16);
eval_fn_t pick_eval_d1_gauss(int mt0, int f16);
eval_fn_t pick_eval_d1_gen(int mt0, int f16);
eval_fn_t pick_eval_d2_cat(int mt0, int f16);
eval_fn_t pick_eval_d2_gauss(int mt0, int f16);
eval_fn_t pick_eval_d3_cat(int mt0, int f16);
eval_fn_t pick_eval_d3_gauss(int mt0, int f16);
// the fast builds (eval_kernel<..., FAST = true>): nullptr where there is none (more than kFastMaxMT0 tiles in layer 0)
eval_fn_t pick_eval_d1_cat_fast(int mt0, int f16);
eval_fn_t pick_eval_d1_gauss_fast(int mt0, int f16);
eval_fn_t pick_eval_d2_cat_fast(int mt0, int f16);
eval_fn_t pick_eval_d2_gauss_fast(int mt0, int f16);
eval_fn_t pick_eval_d3_cat_fast(int mt0, int f16);
eval_fn_t pick_eval_d3_gauss_fast(int mt0, int f16);
}

// lk: likelihood class of the build (npbnn::lik_class); the float64 row-wise class has single-candidate builds only
static eval_fn_t npbnn_pick_eval_kernel(int mt0, int mti, int f16, int n_cand, int lk, bool fast = false, bool blocked = false) {
    using namespace npbnn;
    if (fast) {
        const bool g = lk == kLikGauss;
        if (blocked) f16 = 2;                 // (fast_launch_ok: block structure only on the fp16-split path)
        if (n_cand <= 1) return g ? pick_eval_d1_gauss_fast(mt0, f16) : pick_eval_d1_cat_fast(mt0, f16);
        if (n_cand == 2) return g ? pick_eval_d2_gauss_fast(mt0, f16) : pick_eval_d2_cat_fast(mt0, f16);
        return g ? pick_eval_d3_gauss_fast(mt0, f16) : pick_eval_d3_cat_fast(mt0, f16);
    }
    if (mti != 1) return lk == kLikGen ? pick_eval_mti8_gen(mt0, f16) : lk == kLikGauss ? pick_eval_mti8_gauss(mt0, f16) : pick_eval_mti8_cat(mt0, f16);
    if (lk == kLikGen) return pick_eval_d1_gen(mt0, f16);
    const bool g = lk == kLikGauss;
    if (n_cand <= 1) return g ? pick_eval_d1_gauss(mt0, f16) : pick_eval_d1_cat(mt0, f16);
    if (n_cand == 2) return g ? pick_eval_d2_gauss(mt0, f16) : pick_eval_d2_cat(mt0, f16);
    return g ? pick_eval_d3_gauss(mt0, f16) : pick_eval_d3_cat(mt0, f16);
}

namespace {

thread_local std::string g_last_error;

struct Dataset {
    float* X = nullptr;
    int* labels = nullptr;
    float* targets = nullptr;
    float* inst_w = nullptr;
    int64_t n_rows = 0;
    int n_tiles = 0;
    int F = 0, Fp = 0, k = 0;
    float* X16 = nullptr;      // fp16-split copy (built lazily on the device), row stride Fp16 floats
    int Fp16 = 0;
    int f16_state = 0;         // 0 not built, 1 usable, -1 not representable (inf/NaN or outside the fp16 range), -2 representable but
                               // too coarse for some column: its entries span too many powers of two for a pair of fp16 numbers
    int f16_worst_col = -1;    // column with the largest (max entry error / mean |entry|) of the fp16 pair, and that ratio
    double f16_worst_ratio = 0.0;
    bool borrowed = false;     // X / X16 belong to another ctx (npbnn_share_data)
};

}  // namespace

struct npbnn_ctx {
    int device = 0;
    int n_cu = 256;
    size_t lds_limit = 160 * 1024;
    hipStream_t stream = nullptr;
    std::string err;
    Dataset ds[2];
    double* d_classw = nullptr;
    int n_classw = 0;
    bool arch_set = false;
    npbnn_arch arch{};
    NetMeta net{};
    int n_weights = 0;
    int mt0_template = 1;
    int l0_option = 0;             // NPBNN_L0_AUTO / _F32 / _F16
    int fast_option = 1;           // NPBNN_OPT_FAST_TAILS
    int slopes_option = 0;         // NPBNN_OPT_TRAINABLE_SLOPES: the image holds a slot per hidden layer for the activation slope
    SlopeState* d_slopes = nullptr; // trainable slopes of the device chain (npbnn_chain_cfg.slope_idx ...)
    int* d_sidx = nullptr;          // [slope_cap] pre-drawn slope entries ...
    double* d_sdelta = nullptr;     // ... and steps
    size_t slope_cap = 0;
    bool batch_slopes = false;      // the batch in flight carries slopes (chain_finish reads them back)
    int persist_option = 1;        // NPBNN_OPT_PERSISTENT
    // layer-0 block structure (npbnn_set_layer_mask): which (16-node tile, 16-feature group) blocks of the mask hold a nonzero;
    // empty = dense
    std::vector<unsigned char> l0_blocks;      // [mt][ceil(in_dim / 16)]
    float* d_xscale = nullptr;     // per-feature power-of-two scales of the fp16-split path (from the training matrix)
    float* d_wscale = nullptr;
    int scale_F = 0;
    int* d_overflow = nullptr;
    // parameter blocks of the kernels: device copies (kernels take a pointer) + pinned host staging
    EvalParams* d_eparams = nullptr;
    FinalizeParams* d_fparams = nullptr;
    ChainParams* d_cparams = nullptr;
    char* h_params = nullptr;      // pinned: EvalParams | FinalizeParams | ChainParams
    float* d_w2scale = nullptr;
    // device work buffers
    double* d_wraw = nullptr;      // packed float64 weights
    double* d_colov = nullptr;     // column override (in_dim doubles)
    float* d_image = nullptr;      // float32 fragment image
    int* d_w2img = nullptr;        // packed-weight index -> image float index
    double* d_partials = nullptr;
    int partial_waves = 0;
    unsigned* d_conf = nullptr;    // NPBNN_MAX_WIDTH^2
    npbnn_eval_out* d_out = nullptr;
    float* d_y = nullptr;
    size_t d_y_cap = 0;
    // pinned host staging
    double* h_w = nullptr;
    size_t h_w_cap = 0;
    npbnn_eval_out* h_out = nullptr;
    unsigned* h_conf = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // device-resident chain.  d_res / h_res: one block [ChainDev | overflow flag | W_cur | accepted | logLik' | logPrior'] so that a
    // single copy brings the whole outcome of a batch to the (pinned) host side; d_chain, d_wcur, d_acc, d_llp, d_lpp point into it
    char* d_res = nullptr;
    char* h_res = nullptr;
    size_t res_cap = 0, res_k = 0, res_nw = 0;
    int* d_chain_ovf = nullptr;
    double its_per_pass = 0.0;     // iterations a launch decided on average in the previous batch (0: unknown)
    double accept_rate = -1.0;     // acceptance rate of the previous batch (< 0: unknown)
    double* d_wcur = nullptr;
    double* d_pv = nullptr;        // [kMaxCand][M] proposed values of the candidates in flight
    size_t pv_cap = 0;
    // NPBNN_SCHED_PERSIST_SERIAL (spec_round): outcome-speculative preparation
    SpecState* d_spec = nullptr;
    double* d_spec_pv = nullptr;   // [3][kSpecOutcomes][kMaxCand][M]
    size_t spec_pv_cap = 0;        // M capacity
    unsigned* d_spec_touch = nullptr;   // [kMaxCand][n_weights] touch tables: pass tags (cleared before they could repeat) ...
    double* d_spec_tval = nullptr;      // ... and values
    double* d_spec_prw = nullptr;       // [n_weights] per-weight prior constants of spec_rounds (ChainParams::spec_prior_w)
    std::vector<double> spec_prw_key;   // what d_spec_prw was built from: prior kind, the per-layer scales
    size_t spec_touch_cap = 0;     // weights capacity
    unsigned spec_gen = 0;         // pass tags handed out so far
    double* d_mask = nullptr;
    ChainDev* d_chain = nullptr;
    int* d_idx = nullptr;
    double* d_delta = nullptr;
    int* d_pos = nullptr;
    float* d_pscale = nullptr;
    size_t draw_cap = 0;        // K*M capacity of d_idx / d_delta
    int* d_cnt = nullptr;
    double* d_logu = nullptr;
    unsigned char* d_acc = nullptr;
    double* d_llp = nullptr;
    double* d_lpp = nullptr;
    size_t iter_cap = 0;        // K capacity
    EvalParams* d_gparams = nullptr;   // parameter block of a group pass led by this context (npbnn_chains_run_batched)
    EvalParams* h_gparams = nullptr;   // its page-locked staging twin
    double* d_pscale_w = nullptr;  // [n_weights] per-weight prior scales of the current batch (npbnn_chain_cfg.prior_scale_w)
    double* d_smult = nullptr;  // [K][k_targets] sigma multipliers, [K] Hastings terms (regression with an estimated error parameter)
    double* d_hast = nullptr;
    size_t smult_cap = 0;       // K capacity of the two
    // exchange run (npbnn_chains_run_exchange): [ExchangeParams | swap_j | swap_k | swap_logu || state | records | cold weights]
    char* d_xbuf = nullptr;
    char* h_xbuf = nullptr;
    size_t xbuf_cap = 0;
    hipEvent_t ev_x = nullptr;
    // feature matrices shared between the chains of one run (npbnn_share_data): a borrower points at its owner, an owner
    // counts its borrowers and outlives them (a destroyed owner lingers until the last borrower lets go)
    // flag-ordered overlapped chain schedule: the launches alternate between these two streams
    hipStream_t stream_e[2] = {nullptr, nullptr};
    bool sync_failed = false;      // a wait timed out once: the schedule stays off for this context
    int debug_sync_skip = -1;      // npbnn_debug_sync_skip_ (diagnostics, not part of the ABI)
    npbnn_ctx* data_owner = nullptr;
    int n_borrowers = 0;
    bool zombie = false;
};

namespace {

constexpr double kPersistSerialAccept = 0.07;   // NPBNN_SCHED_AUTO: acceptance rate above which the persistent launch decides between the passes
constexpr size_t kChainMinCapacity = 2048;    // iterations the per-batch chain buffers are sized for at least (allocation is slow)

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

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, NPBNN_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

void free_dataset(Dataset& d) {
    if (d.X && !d.borrowed) (void)hipFree(d.X);
    if (d.labels) (void)hipFree(d.labels);
    if (d.targets) (void)hipFree(d.targets);
    if (d.inst_w) (void)hipFree(d.inst_w);
    if (d.X16 && !d.borrowed) (void)hipFree(d.X16);
    d = Dataset();
}

void destroy_ctx(npbnn_ctx* c);

// a borrower lets go of its owner's matrices (before it uploads its own, or when it is destroyed)
void unshare_data(npbnn_ctx* ctx) {
    npbnn_ctx* owner = ctx->data_owner;
    if (!owner) return;
    for (int w = 0; w < 2; ++w)
        if (ctx->ds[w].borrowed) { ctx->ds[w].X = nullptr; ctx->ds[w].X16 = nullptr; ctx->ds[w].borrowed = false; ctx->ds[w].f16_state = 0; }
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
    for (int l = 0; l < a->n_layers; ++l) {
        const int out = a->out_dim[l];
        if (out < 1 || out > NPBNN_MAX_WIDTH)
            return fail(ctx, NPBNN_E_ARG, "set_arch: layer %d has %d nodes; this backend supports 1..%d per layer", l, out,
                        NPBNN_MAX_WIDTH);
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
            off += slots * (f16 ? 512 : 256);
        } else if (l == 1 && net.l1_f16) {
            off += ((net.L[0].mt + 1) / 2) * 512;      // a high and a low block of 256 floats per K-step
        } else {
            off += L.kt * L.mt * 256;
        }
        L.w_off = woff;
        woff += out * (in + L.has_bias);
        in = out;
    }
    for (int l = 0; l < a->n_layers; ++l) {
        net.L[l].bias_off = off;
        off += 16 * net.L[l].mt;
    }
    if (ctx->n_classw > 0) {            // class weights ride in the image only when there are any
        net.classw_off = off;
        off += NPBNN_MAX_WIDTH;
    } else {
        net.classw_off = -1;
    }
    net.slope_off = -1;
    if (ctx->slopes_option) {           // a slot per hidden layer for the candidates' activation slopes (filled in LDS by a chain pass)
        net.slope_off = off;
        off += kMaxLayers;
    }
    net.image_floats = round_up(off, 64);   // a multiple of 256 B (the LDS copies of several candidates sit back to back)
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

WaveLayout layout_for(const npbnn_ctx* ctx, const Dataset& d, bool predict_only = false) {
    return make_wave_layout(d.labels != nullptr, d.inst_w != nullptr, d.targets ? ctx->net.k_targets : 0, ctx->net.L[0].kt,
                            predict_only ? NPBNN_LIK_NONE : ctx->net.lik_kind);
}

int pick_waves_per_block(const npbnn_ctx* ctx, size_t* lds_bytes, int n_cand, const WaveLayout& lay, bool predict_only = false, bool fast = false) {
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
                       ctx->d_wscale);
    std::vector<unsigned> h((size_t)Fp16);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_max, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_max);
    ctx->scale_F = tr.F;
    tr.f16_state = 0;
    for (unsigned bits : h) {
        float m;
        memcpy(&m, &bits, 4);
        if (!std::isfinite(m)) tr.f16_state = -1;     // inf / NaN in the data: stay on the exact float32 path
    }
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
            const int Fq = d.Fp;
            unsigned* d_err = nullptr;
            unsigned long long* d_sum = nullptr;
            HIP_TRY(ctx, hipMalloc(&d_err, (size_t)Fq * sizeof(unsigned)));
            HIP_TRY(ctx, hipMalloc(&d_sum, (size_t)Fq * sizeof(unsigned long long)));
            HIP_TRY(ctx, hipMemsetAsync(d_err, 0, (size_t)Fq * sizeof(unsigned), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(d_sum, 0, (size_t)Fq * sizeof(unsigned long long), ctx->stream));
            hipLaunchKernelGGL(split_quality_kernel, dim3((Fq + 255) / 256, (unsigned)((d.n_rows + 1023) / 1024)), dim3(256), 0, ctx->stream,
                               (const float*)d.X, (long long)d.n_rows, d.Fp, (const float*)ctx->d_xscale, d_err, d_sum);
            std::vector<unsigned> h_err((size_t)Fq);
            std::vector<unsigned long long> h_sum((size_t)Fq);
            HIP_TRY(ctx, hipMemcpyAsync(h_err.data(), d_err, h_err.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(h_sum.data(), d_sum, h_sum.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(d_err);
            (void)hipFree(d_sum);
            d.f16_worst_col = -1;
            d.f16_worst_ratio = 0.0;
            for (int c = 0; c < d.F; ++c) {
                float e;
                memcpy(&e, &h_err[(size_t)c], 4);
                const double mean_abs = (double)h_sum[(size_t)c] / 4294967296.0 / (double)d.n_rows;
                if (!(mean_abs > 0.0)) continue;                 // an all-zero column is exact
                const double ratio = (double)e / mean_abs;
                if (ratio > d.f16_worst_ratio) { d.f16_worst_ratio = ratio; d.f16_worst_col = c; }
            }
            if (d.f16_worst_ratio > (double)kF16QualityTol && !getenv("NPBNN_F16_NO_QUALITY_CHECK")) d.f16_state = -2;   // heavy-tailed column(s)
        }
        if (d.f16_state < 0 && d.X16) { (void)hipFree(d.X16); d.X16 = nullptr; }      // (nobody will read it)
    }
    *usable = d.f16_state > 0 ? 1 : 0;
    return NPBNN_OK;
}

int rebuild_net(npbnn_ctx* ctx, bool f16);

struct LaunchPlan {
    eval_fn_t fn;
    int n_cand;
    int grid, wpb;
    size_t lds;
    int n_waves;
    bool fast;
};

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
int plan_launch(npbnn_ctx* ctx, int which, LaunchPlan* lp, int force_f32 = 0, int want_cand = 1, bool predict_only = false, bool lik_only = false) {
    Dataset& d = ctx->ds[which];
    bool want_f16 = false;
    if (!force_f32 && ctx->l0_option != NPBNN_L0_F32) {
        int usable = 0;
        int rc0 = ensure_x16(ctx, which, &usable);
        if (rc0) return rc0;
        if (!usable && ctx->l0_option == NPBNN_L0_F16) {
            if (d.f16_state == -2)
                return fail(ctx, NPBNN_E_RANGE, "fp16-split layer 0 was requested but column %d spans too many powers of two for a pair of fp16 "
                                                "numbers (largest entry error %.2e of its mean |value|; bound %.2e)", d.f16_worst_col,
                            d.f16_worst_ratio, (double)kF16QualityTol);
            return fail(ctx, NPBNN_E_RANGE, "fp16-split layer 0 was requested but the data cannot be represented in it");
        }
        want_f16 = usable != 0;
    }
    if ((ctx->net.l0_f16 != 0) != want_f16) {
        int rc0 = rebuild_net(ctx, want_f16);
        if (rc0) return rc0;
    }
    size_t lds = 0;
    // speculative passes: as many candidates as still leave >= 8 waves per workgroup (only the MTI = 1 builds have them)
    int n_cand = (max_inner_tiles(ctx->net) == 1 && (predict_only || !lik_needs_row_scratch(ctx->net.lik_kind))) ? want_cand : 1;
    if (n_cand > kMaxCand) n_cand = kMaxCand;
    const WaveLayout lay = layout_for(ctx, d, predict_only);
    const bool fast = lik_only && !predict_only && fast_launch_ok(ctx, d);
    while (n_cand > 1 && pick_waves_per_block(ctx, &lds, n_cand, lay, predict_only, fast) < 8) --n_cand;
    lp->n_cand = n_cand;
    lp->fast = fast;
    const int wpb = pick_waves_per_block(ctx, &lds, n_cand, lay, predict_only, fast);
    if (wpb == 0)
        return fail(ctx, NPBNN_E_ARG, "network too large: weight image of %d KiB does not fit the %zu KiB LDS of a CU",
                    ctx->net.image_floats * 4 / 1024, ctx->lds_limit / 1024);
    lp->fn = predict_only ? npbnn_pick_eval_kernel(ctx->net.L[0].mt, max_inner_tiles(ctx->net) == 1 ? 1 : 8, ctx->net.l0_f16, n_cand, kLikCat)
                          : npbnn_pick_eval_kernel(ctx->net.L[0].mt, max_inner_tiles(ctx->net) == 1 ? 1 : 8, ctx->net.l0_f16, n_cand,
                                                   lik_class(ctx->net.lik_kind), fast, fast && l0_blocked(ctx->net));
    if (!lp->fn) return fail(ctx, NPBNN_E_STATE, "no evaluation kernel for this shape (internal error)");
    lp->wpb = wpb;
    lp->lds = lds;
    int grid = (d.n_tiles + wpb - 1) / wpb;
    if (grid > ctx->n_cu) grid = ctx->n_cu;     // persistent: one workgroup per CU
    if (grid < 1) grid = 1;
    lp->grid = grid;
    lp->n_waves = grid;            // one partial record per workgroup
    lp->lds = lds + 64;            // (+ the flag word of the device-side waits, behind the images and the rings)
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lp->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp->lds));
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
    const int total = pack_item_count(ctx->net, true);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_wraw, d_co,
                       ctx->n_classw ? ctx->d_classw : nullptr, ctx->d_image, ctx->net,
                       ctx->net.l0_f16 ? ctx->d_wscale : nullptr, ctx->d_overflow);
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

}  // namespace

namespace {
int rebuild_net(npbnn_ctx* ctx, bool f16) {
    int rc = build_net(ctx, &ctx->arch, f16);
    if (rc) return rc;
    if (ctx->d_image) { (void)hipFree(ctx->d_image); ctx->d_image = nullptr; }
    if (ctx->d_w2img) { (void)hipFree(ctx->d_w2img); ctx->d_w2img = nullptr; }
    if (ctx->d_w2scale) { (void)hipFree(ctx->d_w2scale); ctx->d_w2scale = nullptr; }
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
                if (L.has_bias && j == 0) pos = L.bias_off + (L.out_perm ? tile_pos(o) : o);
                else {
                    const int c = in_perm ? tile_pos(j - L.has_bias) : j - L.has_bias;
                    const int mt = o / 16, u = L.out_perm ? tile_pos(o) : o % 16;
                    if (l == 0 && f16) {
                        const int ks = c / 32, kg = (c % 32) / 8, jj = c % 8;
                        if (ks < ctx->net.l0_begin[mt] || ks >= ctx->net.l0_end[mt]) pos = kSkipPos;     // outside the block structure: always 0
                        else {
                            const int slot = ctx->net.l0_base[mt] + ks - ctx->net.l0_begin[mt];
                            const int half_index = 2 * L.frag_off + (((slot * 2) * 64) + kg * 16 + u) * 8 + jj;
                            pos = (int)(0x80000000u | (unsigned)half_index);
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

}  // namespace

extern "C" void npbnn_set_global_error_(const char* msg) { g_last_error = msg ? msg : ""; }

namespace {
void destroy_ctx(npbnn_ctx* c) {
    (void)hipSetDevice(c->device);
    free_dataset(c->ds[0]);
    free_dataset(c->ds[1]);
    if (c->d_classw) (void)hipFree(c->d_classw);
    if (c->d_wraw) (void)hipFree(c->d_wraw);
    if (c->d_colov) (void)hipFree(c->d_colov);
    if (c->d_xscale) (void)hipFree(c->d_xscale);
    if (c->d_wscale) (void)hipFree(c->d_wscale);
    if (c->d_overflow) (void)hipFree(c->d_overflow);
    if (c->d_eparams) (void)hipFree(c->d_eparams);
    if (c->d_fparams) (void)hipFree(c->d_fparams);
    if (c->d_cparams) (void)hipFree(c->d_cparams);
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
    void* chain_bufs[] = {c->d_spec, c->d_spec_pv, c->d_spec_touch, c->d_spec_tval, c->d_spec_prw, c->d_res, c->d_pv, c->d_mask, c->d_idx, c->d_delta, c->d_pos, c->d_pscale, c->d_smult, c->d_hast, c->d_pscale_w, c->d_slopes, c->d_sidx, c->d_sdelta};
    for (void* b : chain_bufs)
        if (b) (void)hipFree(b);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->ev[0]) (void)hipEventDestroy(c->ev[0]);
    if (c->ev[1]) (void)hipEventDestroy(c->ev[1]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
}  // namespace

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
    if (e == hipSuccess) e = hipMalloc(&c->d_conf, (size_t)NPBNN_MAX_WIDTH * NPBNN_MAX_WIDTH * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&c->d_out, sizeof(npbnn_eval_out));
    if (e == hipSuccess) e = hipMalloc(&c->d_overflow, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&c->d_eparams, sizeof(EvalParams));
    if (e == hipSuccess) e = hipMalloc(&c->d_fparams, sizeof(FinalizeParams));
    if (e == hipSuccess) e = hipMalloc(&c->d_cparams, sizeof(ChainParams));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_params, sizeof(EvalParams) + sizeof(FinalizeParams) + sizeof(ChainParams));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_out, sizeof(npbnn_eval_out));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_conf, (size_t)NPBNN_MAX_WIDTH * NPBNN_MAX_WIDTH * sizeof(unsigned));
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
    if (what == NPBNN_INFO_WAVES_PER_BLOCK) { size_t lds = 0; *out = pick_waves_per_block(ctx, &lds, 1, layout_for(ctx, ctx->ds[0])); return NPBNN_OK; }
    if (what == NPBNN_INFO_N_CU) { *out = ctx->n_cu; return NPBNN_OK; }
    if (what == NPBNN_INFO_FAST_TAILS) { *out = (ctx->arch_set && fast_launch_ok(ctx, ctx->ds[0])) ? 1 : 0; return NPBNN_OK; }
    return fail(ctx, NPBNN_E_ARG, "get_info: unknown item %d", what);
}

static int eval_once(npbnn_ctx* ctx, const double* W_packed, const double* act_prm, const double* col_override, double lik_temp,
                     const double* sigma, int which, npbnn_eval_out* out, int64_t* confusion, int force_f32, int* overflowed) {
    const int lik = ctx->net.lik_kind;
    Dataset& d = ctx->ds[which];
    LaunchPlan lp;
    int rc = plan_launch(ctx, which, &lp, force_f32, 1, false, confusion == nullptr);
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
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_conf, 0, (size_t)C * C * sizeof(unsigned), ctx->stream));
        p.confusion = ctx->d_conf;
    }
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
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
    hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
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
            const int total = pack_item_count(ctx->net, true);
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(int), ctx->stream));
            for (int j = 0; j < g; ++j)
                hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_wraw + (size_t)j * wn,
                                   (const double*)nullptr, ctx->n_classw ? ctx->d_classw : nullptr,
                                   ctx->d_image + (size_t)j * ctx->net.image_floats, ctx->net, ctx->net.l0_f16 ? ctx->d_wscale : nullptr,
                                   ctx->d_overflow);
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
            hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
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

extern "C" int npbnn_comm_allgather_inplace_stream_(npbnn_comm* c, double* d_buf, int count, void* stream);
extern "C" int npbnn_comm_info_(const npbnn_comm* c, int* device, int* rank, int* nranks);
extern "C" void npbnn_comm_abort_(npbnn_comm* c);
extern "C" int npbnn_comm_wait_stream_(npbnn_comm* c, void* stream, const char* what);

namespace {

// ---- a device batch of the chain in three phases: prepare (uploads, parameter blocks, first step), enqueue passes, collect ----
// one block, device and page-locked host twin: [ChainDev | overflow | W | cnt | log u || accepted | logLik' | logPrior'];
// everything before `||` goes up in ONE copy at the start of a batch, the whole block comes back in one at its end
struct ResLayout { size_t w, cnt, logu, acc, llp, lpp, total; };
ResLayout res_layout(size_t kc, size_t wb) {
    const auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
    ResLayout L;
    L.w = 512;
    L.cnt = L.w + up256(wb);
    L.logu = L.cnt + up256(kc * sizeof(int));
    L.acc = L.logu + up256(kc * sizeof(double));
    L.llp = L.acc + up256(kc);
    L.lpp = L.llp + up256(kc * sizeof(double));
    L.total = L.lpp + up256(kc * sizeof(double));
    return L;
}

struct ChainBatch {
    LaunchPlan lp;
    ResLayout RL;
    int D = 1, schedule = NPBNN_SCHED_SERIAL, K = 0, M = 0;
    bool overlap = false;
    bool persist = false;                      // persistent form: one launch loops over the passes (device flags order them)
    bool sync = false;                         // overlapped schedule with the launches alternating between two streams (device flags)
    bool forked = false;                       // sync: the two streams have been made to wait for ctx->stream
    size_t wb = 0;
    int launch = 0;                            // launches enqueued so far (overlapped schedule: the pass parity follows it)
    unsigned long long* d_stamps = nullptr;    // diagnostics
    double tw0 = 0.0, tw1 = 0.0;
};

double wall_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// seg_len > 0: the chain stops deciding at iteration seg_len until an exchange kernel moves the limit (exchange run)
int chain_prepare(npbnn_ctx* ctx, const npbnn_chain_cfg* cfg, const double* W_in, const double* mask_packed, int32_t K, int32_t M,
                  const int32_t* idx, const double* delta, const int32_t* cnt, const double* log_u, int seg_len, ChainBatch* B,
                  bool alone_on_device = true, int group_blocks = 0) {
    // alone_on_device: no other chain's launches share the GPU with this batch (the two-stream schedule counts on that)
    // group_blocks > 0: the chain is one of a group pass (npbnn_chains_run_batched): one candidate per launch, overlapped schedule,
    // its sums come from that many evaluating workgroups of the group's launches
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!cfg || !W_in || K < 1 || M < 1 || !idx || !delta || !cnt || !log_u) return fail(ctx, NPBNN_E_ARG, "chain_run: bad arguments");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "chain_run: call npbnn_set_arch first");
    B->tw0 = wall_us();
    const int lik = ctx->net.lik_kind;
    if (lik == NPBNN_LIK_NONE) return fail(ctx, NPBNN_E_STATE, "chain_run: the architecture has no likelihood");
    if (cfg->prior_kind < 0 || cfg->prior_kind > NPBNN_PRIOR_LAPLACE) return fail(ctx, NPBNN_E_ARG, "chain_run: prior_kind=%d", cfg->prior_kind);
    for (int t = 0; t < K; ++t)
        if (cnt[t] < 0 || cnt[t] > M) return fail(ctx, NPBNN_E_ARG, "chain_run: cnt[%d]=%d outside 0..%d", t, cnt[t], M);
    {
        int32_t idx_max = -1;                          // branch-free so that it vectorises: this is K*M entries per batch
        for (size_t i = 0; i < (size_t)K * M; ++i) idx_max = idx[i] > idx_max ? idx[i] : idx_max;
        if (idx_max >= ctx->n_weights) return fail(ctx, NPBNN_E_ARG, "chain_run: weight index %d out of range", idx_max);
    }
    Dataset& d = ctx->ds[0];
    int rc = check_dataset_for_lik(ctx, d, lik);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchPlan& lp = B->lp;
    int want_cand = cfg->n_candidates;
    if (want_cand < 1) want_cand = kMaxCand;            // 0 = as many as fit
    if (group_blocks > 0) want_cand = 1;
    rc = plan_launch(ctx, 0, &lp, cfg->force_f32, want_cand, false, true);
    if (rc) return rc;
    const int D = lp.n_cand;
    // schedule: overlapping the decision of a pass with the evaluation of the next pays as long as most passes reject everything
    int schedule = group_blocks > 0 ? NPBNN_SCHED_OVERLAP : cfg->schedule;
    if (schedule != NPBNN_SCHED_SERIAL && schedule != NPBNN_SCHED_OVERLAP && schedule != NPBNN_SCHED_OVERLAP2 && schedule != NPBNN_SCHED_PERSIST &&
        schedule != NPBNN_SCHED_PERSIST_SERIAL) {
        const double p_acc = ctx->accept_rate < 0 ? 0.0 : ctx->accept_rate;
        // (measured, config-2 shapes, tools/stress_schedules.py 10000 1 2 4: up to 34 % of the proposals accepted - 71 % of the passes -
        // the overlapped forms lead, 48-63 k against 45-50 k it/s; config 4 at 46 % / 84 %: serial 18.8 k against 17.4-18.4 k)
        schedule = (1.0 - std::pow(1.0 - p_acc, D)) < 0.75 ? NPBNN_SCHED_OVERLAP : NPBNN_SCHED_SERIAL;
        // Where overlapping pays and the chain has the GPU to itself, the persistent form of it: one launch whose workgroups loop
        // over the passes (no launch boundary between passes; a workgroup that is through with pass L starts pass L + 1 while
        // others still finish L).  Its device-side waits only need the launch's workgroups resident together - one per compute
        // unit, at most as many as there are - and are bounded: a time-out ends the batch with NPBNN_E_SYNC, state untouched, and
        // the context stays on kernel boundaries from then on.  (The two-stream form, NPBNN_SCHED_OVERLAP2, is never picked here:
        // it also needs the two streams on hardware queues of their own, which nothing promises.)
        if (schedule == NPBNN_SCHED_OVERLAP && alone_on_device && seg_len == 0 && !ctx->sync_failed && ctx->persist_option)
            schedule = NPBNN_SCHED_PERSIST;
        // A chain that moves: in the overlapped forms every pass that accepts something voids the pass in flight behind it (at 28 %
        // acceptance 63 % of the passes do), and on kernel boundaries a decision between two passes costs a step kernel and two
        // boundaries.  The persistent launch with the decision between the passes (NPBNN_SCHED_PERSIST_SERIAL) wastes no pass and
        // decides in a few microseconds: above kPersistSerialAccept of the proposals accepted it leads (measured).
        if (alone_on_device && seg_len == 0 && !ctx->sync_failed && ctx->persist_option && p_acc > kPersistSerialAccept && group_blocks == 0 &&
            !cfg->slope_idx)
            schedule = NPBNN_SCHED_PERSIST_SERIAL;
    }
    if ((schedule == NPBNN_SCHED_OVERLAP2 || schedule == NPBNN_SCHED_PERSIST) && ctx->sync_failed) schedule = NPBNN_SCHED_OVERLAP;
    if (schedule == NPBNN_SCHED_PERSIST_SERIAL && (ctx->sync_failed || !alone_on_device || seg_len > 0 || group_blocks > 0)) schedule = NPBNN_SCHED_SERIAL;
    // the persistent form needs every workgroup of its launch resident at once: one per compute unit at most, the GPU to itself, and
    // a plain run (an exchange run's kernels go between the passes)
    // (its grid is at most one workgroup per compute unit: the evaluating workgroups are capped at n_cu - 1 below, plus the step's)
    if (schedule == NPBNN_SCHED_PERSIST && (!alone_on_device || seg_len > 0)) schedule = NPBNN_SCHED_OVERLAP;
    const bool pserial = schedule == NPBNN_SCHED_PERSIST_SERIAL;
    const bool persist = schedule == NPBNN_SCHED_PERSIST || pserial;
    const bool overlap = schedule == NPBNN_SCHED_OVERLAP || schedule == NPBNN_SCHED_OVERLAP2 || persist;
    const bool sync = (schedule == NPBNN_SCHED_OVERLAP2 && alone_on_device) || persist;   // (several chains on one GPU: one stream each)
    if (schedule == NPBNN_SCHED_OVERLAP2 && !sync) schedule = NPBNN_SCHED_OVERLAP;
    if (!ctx->stream_e[0]) {      // (with the chain's first batch, whatever its schedule: creating a stream takes milliseconds)
        for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream_e[i], hipStreamNonBlocking));
    }
    if (overlap) {                      // one workgroup of the launch runs the step: the others share the tiles
        int g = lp.grid;
        if (g > ctx->n_cu - 1) g = ctx->n_cu - 1;
        if (g < 1) g = 1;
        lp.grid = g;
        lp.n_waves = g;
    }
    if (group_blocks > 0) { lp.grid = group_blocks; lp.n_waves = group_blocks; }
    rc = ensure_work_buffers(ctx, lp.n_waves);
    if (rc) return rc;
    const size_t wb = (size_t)ctx->n_weights * sizeof(double);
    if ((size_t)K > ctx->res_k || (size_t)ctx->n_weights != ctx->res_nw) {
        size_t kc = (size_t)K > ctx->res_k ? (size_t)K : ctx->res_k;
        if (kc < kChainMinCapacity) kc = kChainMinCapacity;
        const ResLayout L = res_layout(kc, wb);
        if (ctx->d_res) (void)hipFree(ctx->d_res);
        if (ctx->h_res) (void)hipHostFree(ctx->h_res);
        if (ctx->d_mask) (void)hipFree(ctx->d_mask);        // sized by the number of weights as well
        ctx->d_mask = nullptr;
        ctx->d_res = nullptr; ctx->h_res = nullptr; ctx->res_cap = 0; ctx->res_k = 0; ctx->res_nw = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_res, L.total));
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_res, L.total));
        ctx->res_cap = L.total; ctx->res_k = kc; ctx->res_nw = (size_t)ctx->n_weights;
        char* b = ctx->d_res;
        ctx->d_chain = reinterpret_cast<ChainDev*>(b);
        ctx->d_chain_ovf = reinterpret_cast<int*>(b + 448);
        ctx->d_wcur = reinterpret_cast<double*>(b + L.w);
        ctx->d_cnt = reinterpret_cast<int*>(b + L.cnt);
        ctx->d_logu = reinterpret_cast<double*>(b + L.logu);
        ctx->d_acc = reinterpret_cast<unsigned char*>(b + L.acc);
        ctx->d_llp = reinterpret_cast<double*>(b + L.llp);
        ctx->d_lpp = reinterpret_cast<double*>(b + L.lpp);
    }
    const ResLayout RL = res_layout(ctx->res_k, wb);
    static_assert(sizeof(ChainDev) <= 448, "ChainDev must fit its slot of the result block");
    if (mask_packed && !ctx->d_mask) HIP_TRY(ctx, hipMalloc(&ctx->d_mask, wb));
    if ((size_t)M > ctx->pv_cap) {
        if (ctx->d_pv) (void)hipFree(ctx->d_pv);
        ctx->d_pv = nullptr; ctx->pv_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_pv, (size_t)2 * kMaxCand * M * sizeof(double)));
        ctx->pv_cap = (size_t)M;
    }
    const bool spec = pserial && !cfg->slope_idx && !getenv("NPBNN_NO_SPEC_STEP");      // prepare the next pass ahead for every outcome
    if (spec) {
        if (!ctx->d_spec) HIP_TRY(ctx, hipMalloc(&ctx->d_spec, sizeof(SpecState)));
        if ((size_t)M > ctx->spec_pv_cap) {
            if (ctx->d_spec_pv) (void)hipFree(ctx->d_spec_pv);
            ctx->d_spec_pv = nullptr; ctx->spec_pv_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_spec_pv, (size_t)3 * kSpecOutcomes * kMaxCand * M * sizeof(double)));
            ctx->spec_pv_cap = (size_t)M;
        }
        if ((size_t)ctx->n_weights > ctx->spec_touch_cap) {
            if (ctx->d_spec_touch) (void)hipFree(ctx->d_spec_touch);
            if (ctx->d_spec_tval) (void)hipFree(ctx->d_spec_tval);
            if (ctx->d_spec_prw) (void)hipFree(ctx->d_spec_prw);
            ctx->d_spec_touch = nullptr; ctx->d_spec_tval = nullptr; ctx->d_spec_prw = nullptr; ctx->spec_touch_cap = 0;
            ctx->spec_prw_key.clear();
            HIP_TRY(ctx, hipMalloc(&ctx->d_spec_prw, (size_t)ctx->n_weights * sizeof(double)));
            HIP_TRY(ctx, hipMalloc(&ctx->d_spec_touch, (size_t)kMaxCand * ctx->n_weights * 4 * sizeof(unsigned)));
            ctx->spec_touch_cap = (size_t)ctx->n_weights;
            ctx->spec_gen = 0xf0000000u;         // (forces the clearing below)
        }
        if (ctx->spec_gen + (unsigned)K + 8u >= 0xf0000000u) {   // the batch's pass tags (one per pass, at most K + 1 passes) could repeat
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_spec_touch, 0, (size_t)kMaxCand * ctx->spec_touch_cap * 4 * sizeof(unsigned), ctx->stream));
            ctx->spec_gen = 0;
        }
    }
    const size_t need = (size_t)K * M;
    if (need > ctx->draw_cap) {
        const size_t cap = need > (size_t)kChainMinCapacity * M ? need : (size_t)kChainMinCapacity * M;
        if (ctx->d_idx) (void)hipFree(ctx->d_idx);
        if (ctx->d_delta) (void)hipFree(ctx->d_delta);
        if (ctx->d_pos) (void)hipFree(ctx->d_pos);
        if (ctx->d_pscale) (void)hipFree(ctx->d_pscale);
        ctx->d_idx = nullptr; ctx->d_delta = nullptr; ctx->d_pos = nullptr; ctx->d_pscale = nullptr; ctx->draw_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_idx, cap * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_delta, cap * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_pos, cap * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_pscale, cap * sizeof(float)));
        ctx->draw_cap = cap;
    }
    hipStream_t st = ctx->stream;
    if (mask_packed) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mask, mask_packed, wb, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_idx, idx, need * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_delta, delta, need * sizeof(double), hipMemcpyHostToDevice, st));
    const bool f16 = ctx->net.l0_f16 != 0;
    // image position (and fp16-split scale) of every drawn entry, so no kernel needs a dependent lookup
    hipLaunchKernelGGL(gather_pos_kernel, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, st, (const int*)ctx->d_idx, (long long)need,
                       (const int*)ctx->d_w2img, (const float*)(f16 ? ctx->d_w2scale : nullptr), ctx->d_pos,
                       f16 ? ctx->d_pscale : (float*)nullptr);
    ChainDev init{};
    init.logLik = cfg->cur_loglik;
    init.logPrior = cfg->cur_logprior;
    init.logPrior_rep = cfg->cur_logprior;
    for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) init.sigma[j] = cfg->cur_sigma[j];
    init.t = 0;
    init.n_accepted = 0;
    init.n_passes = 0;
    init.void_launch = -2;
    init.n_void = 0;
    init.seg_end = (seg_len > 0 && seg_len < K) ? seg_len : K;
    init.temperature = cfg->temperature;
    init.seg_idx = 0;
    init.poisoned = 0;
    init.prepared = -1;             // (the first step kernel raises it to 0)
    init.aborted = 0;
    init.started = -1;
    init.exchanged = 0;
    for (int i = 0; i < 4; ++i) init.done[i] = 0;
    {   // initial chain state, overflow flag, weights and the per-iteration scalars travel together (head of the block)
        memset(ctx->h_res, 0, 512);
        memcpy(ctx->h_res, &init, sizeof(ChainDev));
        memcpy(ctx->h_res + RL.w, W_in, wb);
        memcpy(ctx->h_res + RL.cnt, cnt, (size_t)K * sizeof(int));
        memcpy(ctx->h_res + RL.logu, log_u, (size_t)K * sizeof(double));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_res, ctx->h_res, RL.acc, hipMemcpyHostToDevice, st));
    }
    for (int l = 0; l < kMaxLayers; ++l) ctx->net.act_prm[l] = 0.f;
    {   // weight image of the current state; accepted candidates are committed to it entry by entry
        const int total = pack_item_count(ctx->net, true);
        hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ctx->d_wcur, (const double*)nullptr,
                           ctx->n_classw ? ctx->d_classw : nullptr, ctx->d_image, ctx->net,
                           f16 ? ctx->d_wscale : nullptr, ctx->d_chain_ovf);
    }
    ChainParams c{};
    c.st = ctx->d_chain;
    c.pass = reinterpret_cast<PassDesc*>(reinterpret_cast<char*>(ctx->d_eparams) + offsetof(EvalParams, pass_desc));   // inside the evaluation's block
    c.w_cur = ctx->d_wcur;
    c.mask = mask_packed ? ctx->d_mask : nullptr;
    c.idx = ctx->d_idx;
    c.delta = ctx->d_delta;
    c.cnt = ctx->d_cnt;
    c.log_u = ctx->d_logu;
    c.hastings = nullptr;
    c.sigma_mult = nullptr;
    if (cfg->sigma_mult || cfg->hastings) {
        if (!cfg->sigma_mult || !cfg->hastings || lik != NPBNN_LIK_GAUSS)
            return fail(ctx, NPBNN_E_ARG, "chain_run: sigma_mult and hastings go together, with the Gaussian likelihood");
        const int kt = ctx->net.k_targets;
        if ((size_t)K > ctx->smult_cap) {
            if (ctx->d_smult) (void)hipFree(ctx->d_smult);
            if (ctx->d_hast) (void)hipFree(ctx->d_hast);
            ctx->d_smult = nullptr; ctx->d_hast = nullptr; ctx->smult_cap = 0;
            const size_t cap = (size_t)K > kChainMinCapacity ? (size_t)K : kChainMinCapacity;
            HIP_TRY(ctx, hipMalloc(&ctx->d_smult, cap * NPBNN_MAX_TARGETS * sizeof(double)));
            HIP_TRY(ctx, hipMalloc(&ctx->d_hast, cap * sizeof(double)));
            ctx->smult_cap = cap;
        }
        for (size_t i = 0; i < (size_t)K * kt; ++i)
            if (!(cfg->sigma_mult[i] > 0.0)) return fail(ctx, NPBNN_E_ARG, "chain_run: sigma_mult[%zu] is not positive", i);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_smult, cfg->sigma_mult, (size_t)K * kt * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hast, cfg->hastings, (size_t)K * sizeof(double), hipMemcpyHostToDevice, st));
        c.sigma_mult = ctx->d_smult;
        c.hastings = ctx->d_hast;
    }
    c.out_acc = ctx->d_acc;
    c.out_ll = ctx->d_llp;
    c.out_lp = ctx->d_lpp;
    c.partials = ctx->d_partials;
    c.image = ctx->d_image;
    c.pos = ctx->d_pos;
    c.pscale = f16 ? ctx->d_pscale : nullptr;
    c.pv = spec ? ctx->d_spec_pv : ctx->d_pv;      // (spec: the first step writes pass 0 into slot (parity 0, outcome 0))
    c.spec = nullptr;
    c.spec_pv = nullptr;
    c.spec_touch = nullptr;
    c.spec_touch_val = nullptr;
    c.spec_prior_w = nullptr;
    c.n_weights_spec = ctx->n_weights;
    c.spec_gen = 0;
    if (spec) {
        c.spec = ctx->d_spec;
        c.spec_pv = ctx->d_spec_pv;
        c.spec_touch = ctx->d_spec_touch;
        c.spec_touch_val = ctx->d_spec_tval;
        c.spec_gen = (int)ctx->spec_gen;
        ctx->spec_gen += (unsigned)K + 8u;             // (a pass decides at least one iteration)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_spec, 0, sizeof(SpecState), st));
    }
    c.overflow = ctx->d_chain_ovf;
    B->d_stamps = nullptr;
    if (getenv("NPBNN_STEP_STAMPS")) {      // diagnostics: per-phase wall-clock stamps of the step kernel
        HIP_TRY(ctx, hipMalloc(&B->d_stamps, 1024 * 8 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemset(B->d_stamps, 0, 1024 * 8 * sizeof(unsigned long long)));
    }
    c.stamps = B->d_stamps;
    c.K = K;
    c.M = M;
    c.D = D;
    c.n_blocks = lp.n_waves;
    c.stop_on_overflow = seg_len > 0 ? 1 : 0;
    c.sync_test_skip = -1;
    if (sync && ctx->debug_sync_skip >= 0) {       // (npbnn_debug_sync_skip_: provoke the time-out of the two-stream schedule, once)
        c.sync_test_skip = ctx->debug_sync_skip;
        ctx->debug_sync_skip = -1;
    }
    c.prior_kind = cfg->prior_kind;
    for (int l = 0; l < kMaxLayers; ++l) {
        c.prior_scale[l] = cfg->prior_scale[l];
        c.half_inv_s2[l] = cfg->prior_scale[l] > 0 ? 0.5 / (cfg->prior_scale[l] * cfg->prior_scale[l]) : 0.0;
    }
    c.prior_scale_w = nullptr;
    if (cfg->prior_scale_w && cfg->prior_kind != NPBNN_PRIOR_UNIFORM) {
        for (int i = 0; i < ctx->n_weights; ++i)
            if (!(cfg->prior_scale_w[i] > 0.0)) return fail(ctx, NPBNN_E_ARG, "chain_run: prior_scale_w[%d] is not positive", i);
        if (!ctx->d_pscale_w) HIP_TRY(ctx, hipMalloc(&ctx->d_pscale_w, wb));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pscale_w, cfg->prior_scale_w, wb, hipMemcpyHostToDevice, st));
        c.prior_scale_w = ctx->d_pscale_w;
    }
    if (spec) {
        c.spec_prior_w = c.prior_scale_w;               // a scale per weight (already uploaded), or:
        if (!c.prior_scale_w) {
            std::vector<double> key{(double)cfg->prior_kind};
            for (int l = 0; l < ctx->net.n_layers; ++l) key.push_back(cfg->prior_scale[l]);
            if (key != ctx->spec_prw_key) {             // (changes with a Gibbs step of the scales only)
                std::vector<double> prw((size_t)ctx->n_weights, 0.0);
                for (int l = 0; l < ctx->net.n_layers; ++l) {
                    const LayerMeta& L = ctx->net.L[l];
                    const double v = cfg->prior_kind == NPBNN_PRIOR_NORMAL ? c.half_inv_s2[l] : cfg->prior_scale[l];
                    for (int i = 0; i < L.out_dim * (L.in_dim + L.has_bias); ++i) prw[(size_t)L.w_off + i] = v;
                }
                HIP_TRY(ctx, hipMemcpy(ctx->d_spec_prw, prw.data(), prw.size() * sizeof(double), hipMemcpyHostToDevice));
                ctx->spec_prw_key = key;
            }
            c.spec_prior_w = ctx->d_spec_prw;
        }
    }
    c.slopes = nullptr;
    c.slope_idx = nullptr;
    c.slope_delta = nullptr;
    c.n_slopes = 0;
    c.slope_term_in = 0;
    ctx->batch_slopes = false;
    if (cfg->slope_idx || cfg->slope_delta) {          // trainable activation slopes
        if (!cfg->slope_idx || !cfg->slope_delta || cfg->n_slopes < 1 || cfg->n_slopes > kMaxLayers || cfg->n_slopes != ctx->net.n_layers - 1)
            return fail(ctx, NPBNN_E_ARG, "chain_run: slope_idx and slope_delta go together, with one slope per hidden layer (got %d for %d layers)",
                        cfg->n_slopes, ctx->net.n_layers);
        if (ctx->net.slope_off < 0) return fail(ctx, NPBNN_E_STATE, "chain_run: trainable slopes need NPBNN_OPT_TRAINABLE_SLOPES");
        if (seg_len > 0 || group_blocks > 0) return fail(ctx, NPBNN_E_ARG, "chain_run: trainable slopes run in plain batches only");
        for (int t = 0; t < K; ++t)
            if (cfg->slope_idx[t] < 0 || cfg->slope_idx[t] >= cfg->n_slopes) return fail(ctx, NPBNN_E_ARG, "chain_run: slope_idx[%d] out of range", t);
        if ((size_t)K > ctx->slope_cap) {
            if (ctx->d_sidx) (void)hipFree(ctx->d_sidx);
            if (ctx->d_sdelta) (void)hipFree(ctx->d_sdelta);
            ctx->d_sidx = nullptr; ctx->d_sdelta = nullptr; ctx->slope_cap = 0;
            const size_t cap = (size_t)K > kChainMinCapacity ? (size_t)K : kChainMinCapacity;
            HIP_TRY(ctx, hipMalloc(&ctx->d_sidx, cap * sizeof(int)));
            HIP_TRY(ctx, hipMalloc(&ctx->d_sdelta, cap * sizeof(double)));
            ctx->slope_cap = cap;
        }
        if (!ctx->d_slopes) HIP_TRY(ctx, hipMalloc(&ctx->d_slopes, sizeof(SlopeState)));
        SlopeState init_s{};
        for (int l = 0; l < cfg->n_slopes; ++l) init_s.cur[l] = cfg->cur_slopes[l];
        // (pageable sources: the copies are staged by the runtime before the calls return)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slopes, &init_s, sizeof(SlopeState), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sidx, cfg->slope_idx, (size_t)K * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_sdelta, cfg->slope_delta, (size_t)K * sizeof(double), hipMemcpyHostToDevice, st));
        c.slopes = ctx->d_slopes;
        c.slope_idx = ctx->d_sidx;
        c.slope_delta = ctx->d_sdelta;
        c.n_slopes = cfg->n_slopes;
        c.slope_term_in = cfg->slope_term_in_prior ? 1 : 0;
        ctx->batch_slopes = true;
    }
    c.w_bound = cfg->w_bound;
    c.lik_temp = cfg->lik_temp;
    c.sigma_given = cfg->sigma_given;
    for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) c.sigma_fixed[j] = cfg->sigma[j];
    c.n_rows = d.n_rows;
    c.net = ctx->net;
    EvalParams p = make_params(ctx, d);
    p.partials = ctx->d_partials;
    p.inst_w = d.inst_w;
    p.use_classw = ctx->n_classw > 0 ? 1 : 0;
    p.has_pass = 1;
    p.pv = spec ? ctx->d_spec_pv : ctx->d_pv;
    p.pos = ctx->d_pos;
    p.pscale = f16 ? ctx->d_pscale : nullptr;
    p.M = M;
    p.chain = overlap ? ctx->d_cparams : nullptr;
    p.sync_mode = spec ? 3 : pserial ? 2 : sync ? 1 : 0;
    p.cand_slopes = c.slopes ? &ctx->d_slopes->cand[0][0][0] : nullptr;
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
    rc = push_chain_params(ctx, c);
    if (rc) return rc;
    // step (prepare candidates) -> [eval -> step (decide + prepare)]* ; a pass consumes 1..D iterations, so the number of
    // passes is only known on the device
    hipLaunchKernelGGL(chain_step_kernel, dim3(1), dim3(1024), 0, st, (const ChainParams*)ctx->d_cparams, 1);
    B->RL = RL;
    B->persist = persist;
    B->sync = sync && !persist;              // (two launch streams to fork and join)
    B->forked = false;
    B->D = D;
    B->schedule = schedule;
    B->overlap = overlap;
    B->K = K;
    B->M = M;
    B->wb = wb;
    B->launch = 0;
    B->tw1 = wall_us();
    return NPBNN_OK;
}

// passes to launch for `rem` iterations: a pass decides between 1 and D of them; what the previous batch's average says is
// needed plus a margin (passes launched after the last iteration return at once)
int passes_for(const npbnn_ctx* ctx, const ChainBatch& B, int rem, double slack) {
    int n = (rem + B.D - 1) / B.D;
    if (B.persist) {
        // the persistent launch ends by itself at the chain's terminal pass, so a generous bound costs nothing - and a second round
        // (results back, look, launch again) costs a host round trip: the worst case, every iteration accepted (one iteration per
        // pass and a void pass after each)
        return 2 * rem + 4;
    }
    if (ctx->its_per_pass >= 1.0) {
        const int est = (int)std::ceil(slack * (double)rem / ctx->its_per_pass);
        if (est > n) n = est;
        // a launch past the end of the batch returns at once (a few microseconds); coming back short costs a host round trip and a
        // second round: lean towards the former
        n += 3 + n / 8;
    }
    if (B.overlap) n += 1;                   // the last pass is decided by the launch after it
    return n;
}

// the same for a segment of an exchange run, where falling short is expensive (no second round): with no history, the
// worst case (every iteration accepted: one iteration per pass, and in the overlapped schedule a void pass after each)
int passes_for_segment(const npbnn_ctx* ctx, const ChainBatch& B, int seg_len, double slack) {
    double est = ctx->its_per_pass >= 1.0 ? (double)seg_len / ctx->its_per_pass : (double)seg_len * (B.overlap ? 2.0 : 1.0);
    const double least = (double)((seg_len + B.D - 1) / B.D);
    if (est < least) est = least;
    return (int)std::ceil(slack * est) + 2 + (B.overlap ? 1 : 0);
}

// flag-ordered overlapped schedule: the launches go to two streams in turn.  Work enqueued on ctx->stream (uploads, the first
// step) must be complete before their first launch (fork), theirs before ctx->stream copies results back (join): both by
// host synchronisation - once a stream has waited for another stream's event, this runtime runs every later launch of the two
// one after the other, which is the very thing the schedule is there to avoid (measured: no overlap at all with event waits)
int chain_fork(npbnn_ctx* ctx, ChainBatch& B) {
    if (!B.sync || B.forked) return NPBNN_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    B.forked = true;
    return NPBNN_OK;
}
int chain_join(npbnn_ctx* ctx, ChainBatch& B) {
    if (!B.sync || !B.forked) return NPBNN_OK;
    for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_e[i]));
    B.forked = false;
    return NPBNN_OK;
}

int chain_enqueue(npbnn_ctx* ctx, ChainBatch& B, int n) {
    hipStream_t st = ctx->stream;
    const LaunchPlan& lp = B.lp;
    if (B.persist) {            // ONE launch stands for the n passes: its workgroups loop over them (eval_kernel, n_loop)
        if (n > 0) hipLaunchKernelGGL(lp.fn, dim3(lp.grid + 1), dim3(lp.wpb * 64), lp.lds, st, (const EvalParams*)ctx->d_eparams, B.launch, n);
        B.launch += n;
    } else if (B.sync) {
        int rc = chain_fork(ctx, B);
        if (rc) return rc;
        for (int i = 0; i < n; ++i, ++B.launch) {    // (bit 30: not the last launch of this round - see sync_step_leave)
            if (i == 1) hipLaunchKernelGGL(sync_gate_kernel, dim3(1), dim3(64), 0, ctx->stream_e[B.launch & 1], ctx->d_chain, B.launch - 1);
            hipLaunchKernelGGL(lp.fn, dim3(lp.grid + 1), dim3(lp.wpb * 64), lp.lds, ctx->stream_e[B.launch & 1], (const EvalParams*)ctx->d_eparams,
                               B.launch | (i + 1 < n ? (1 << 30) : 0), 1);
        }
    } else if (B.overlap) {
        for (int i = 0; i < n; ++i, ++B.launch)
            hipLaunchKernelGGL(lp.fn, dim3(lp.grid + 1), dim3(lp.wpb * 64), lp.lds, st, (const EvalParams*)ctx->d_eparams, B.launch, 1);
    } else {
        for (int i = 0; i < n; ++i, ++B.launch) {
            hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, st, (const EvalParams*)ctx->d_eparams, 0, 1);
            hipLaunchKernelGGL(chain_step_kernel, dim3(1), dim3(1024), 0, st, (const ChainParams*)ctx->d_cparams, 0);
        }
    }
    return NPBNN_OK;
}

// after the result block has come back (h_res): hand the first k_take iterations' outcome to the caller
int chain_finish(npbnn_ctx* ctx, ChainBatch& B, const npbnn_chain_cfg* cfg, double* W_inout, uint8_t* out_accepted, double* out_loglik_prop,
                 double* out_logprior_prop, npbnn_chain_result* result, int k_take, bool exchange_run = false) {
    const ChainDev fin = *reinterpret_cast<const ChainDev*>(ctx->h_res);
    if (fin.n_passes + fin.n_void > 0 && k_take > 0) ctx->its_per_pass = (double)k_take / (fin.n_passes + fin.n_void);
    if (k_take > 0) ctx->accept_rate = (double)fin.n_accepted / k_take;
    if (B.d_stamps) {
        std::vector<unsigned long long> hs(1024 * 8);
        (void)hipMemcpy(hs.data(), B.d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(B.d_stamps);
        B.d_stamps = nullptr;
        double acc[8] = {0};
        int n = 0;
        for (int r = 1; r < 1024; ++r) {
            const unsigned long long* q = &hs[(size_t)r * 8];
            if (!q[0] || !q[6]) continue;
            for (int k = 1; k <= 6; ++k) acc[k] += (double)(q[k] - q[k - 1]) * 0.01;   // 100 MHz wall clock -> us
            ++n;
        }
        if (n) fprintf(stderr, "[npbnn step stamps] prefetch %.2f  reduce %.2f  decide %.2f  commit %.2f  prepare %.2f  finish %.2f us (mean of %d)\n",
                       acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, n);
    }
    const int flags = *reinterpret_cast<const int*>(ctx->h_res + 448);
    if (flags & kFlagStructure) return fail(ctx, NPBNN_E_ARG, "chain_run: a layer-0 weight is not zero where the mask given to npbnn_set_layer_mask is");
    const int overflow = (ctx->net.l0_f16 && (flags & kFlagF16Range)) ? 1 : 0;
    if (overflow && !exchange_run)           // W_inout untouched: the caller re-runs this batch with cfg->force_f32 = 1
        return fail(ctx, NPBNN_E_RANGE, "chain_run: a layer-0 weight left the fp16 range during this batch");
    {
        const char* b = ctx->h_res;
        memcpy(W_inout, b + B.RL.w, B.wb);
        memcpy(out_accepted, b + B.RL.acc, (size_t)k_take);
        if (out_loglik_prop) memcpy(out_loglik_prop, b + B.RL.llp, (size_t)k_take * sizeof(double));
        if (out_logprior_prop) memcpy(out_logprior_prop, b + B.RL.lpp, (size_t)k_take * sizeof(double));
    }
    for (int l = 0; l < NPBNN_MAX_LAYERS; ++l) result->slopes[l] = 0.0;
    if (ctx->batch_slopes) {             // (the stream is idle: the result block has just come back)
        SlopeState fs;
        HIP_TRY(ctx, hipMemcpy(&fs, ctx->d_slopes, sizeof(SlopeState), hipMemcpyDeviceToHost));
        for (int l = 0; l < kMaxLayers && l < NPBNN_MAX_LAYERS; ++l) result->slopes[l] = fs.cur[l];
    }
    result->loglik = fin.logLik;
    result->logprior = fin.n_accepted > 0 ? fin.logPrior_rep : cfg->cur_logprior;
    for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) result->sigma[j] = fin.sigma[j];
    result->n_accepted = fin.n_accepted;
    result->n_passes = fin.n_passes;
    result->n_candidates = B.D;
    result->n_void_passes = fin.n_void;
    result->schedule = B.schedule;
    result->temperature = fin.temperature;
    result->iterations_done = k_take;
    result->overflow = overflow;          // (exchange run: the chain stopped before the proposal that overflows)
    return NPBNN_OK;
}

}  // namespace

extern "C" {

int npbnn_chain_run(npbnn_ctx* ctx, const npbnn_chain_cfg* cfg, double* W_inout, const double* mask_packed, int32_t K, int32_t M,
                    const int32_t* idx, const double* delta, const int32_t* cnt, const double* log_u, uint8_t* out_accepted,
                    double* out_loglik_prop, double* out_logprior_prop, npbnn_chain_result* result) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!result || !out_accepted) return fail(ctx, NPBNN_E_ARG, "chain_run: bad arguments");
    static const bool timing = getenv("NPBNN_CHAIN_TIMING") != nullptr;     // diagnostics: host wall clock per phase
    ChainBatch B;
    int rc = chain_prepare(ctx, cfg, W_inout, mask_packed, K, M, idx, delta, cnt, log_u, 0, &B);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    int t_done = 0, n_rounds = 0;
    const ChainDev* now = reinterpret_cast<const ChainDev*>(ctx->h_res);
    double t_enq = 0.0, t_wait = 0.0;
    while (t_done < K) {       // launch the least number of passes that can finish, look at the counter, repeat if short
        ++n_rounds;
        const double ta = timing ? wall_us() : 0.0;
        const int n_launch = passes_for(ctx, B, K - t_done, 1.0);
        rc = chain_enqueue(ctx, B, n_launch);
        if (!rc) rc = chain_join(ctx, B);
        if (rc) return rc;
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_res, ctx->d_res, B.RL.total, hipMemcpyDeviceToHost, st));   // state + results, one copy
        const double tb = timing ? wall_us() : 0.0;
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (timing) {
            const double tc = wall_us();
            t_enq += tb - ta;
            t_wait += tc - tb;
            fprintf(stderr, "[npbnn chain timing]   round %d: %d launches enqueued in %.0f us, waited %.0f us, t=%d of %d\n", n_rounds, n_launch, tb - ta,
                    tc - tb, reinterpret_cast<const ChainDev*>(ctx->h_res)->t, K);
        }
        if (now->aborted) {         // a device-side wait of the flag-ordered schedule timed out: nothing was decided after it
            ctx->sync_failed = true;
            return fail(ctx, NPBNN_E_SYNC, "chain_run: the flag-ordered overlapped schedule timed out at t=%d; retry on one stream", now->t);
        }
        if (now->t < t_done || (now->t == t_done && !B.overlap))
            return fail(ctx, NPBNN_E_STATE, "chain_run: the device chain made no progress (t=%d)", now->t);
        if (now->t == t_done && n_rounds > 64) return fail(ctx, NPBNN_E_STATE, "chain_run: the device chain is stuck at t=%d", now->t);
        t_done = now->t;
    }
    const double tw2 = wall_us();
    rc = chain_finish(ctx, B, cfg, W_inout, out_accepted, out_loglik_prop, out_logprior_prop, result, K);
    if (rc) return rc;
    if (timing && ctx->d_spec && B.schedule == NPBNN_SCHED_PERSIST_SERIAL) {
        SpecState hs;
        if (hipMemcpy(&hs, ctx->d_spec, sizeof hs, hipMemcpyDeviceToHost) == hipSuccess && hs.rounds > 0)
            fprintf(stderr, "[npbnn chain timing]   step rounds %d: touch %.2f  candidates %.2f  descriptors %.2f  wait for the pass %.2f  decide+publish %.2f  "
                            "commit %.2f us per round\n", hs.rounds, hs.ticks[0] * 0.01 / hs.rounds, hs.ticks[1] * 0.01 / hs.rounds, hs.ticks[2] * 0.01 / hs.rounds,
                    hs.ticks[3] * 0.01 / hs.rounds, hs.ticks[4] * 0.01 / hs.rounds, hs.ticks[5] * 0.01 / hs.rounds);
    }
    if (timing)
        fprintf(stderr, "[npbnn chain timing] K=%d passes=%d (+%d void, %s) rounds=%d: setup %.0f us, passes %.0f us (%.2f us/pass), results %.0f us\n", K,
                result->n_passes, result->n_void_passes, B.overlap ? "overlapped" : "serial", n_rounds, B.tw1 - B.tw0, tw2 - B.tw1,
                (tw2 - B.tw1) / (result->n_passes > 0 ? result->n_passes : 1), wall_us() - tw2);
    return NPBNN_OK;
}


int npbnn_chains_run_batched(npbnn_chain_job* jobs, int32_t n_jobs, int32_t K) {
    if (!jobs || n_jobs < 2 || n_jobs > kMaxCand || K < 1) return fail(nullptr, NPBNN_E_ARG, "chains_run_batched: 2..%d chains, K >= 1", kMaxCand);
    npbnn_ctx* ctx0 = jobs[0].ctx;
    if (!ctx0) return fail(nullptr, NPBNN_E_ARG, "chains_run_batched: job 0 has no context");
    int force_f32 = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const npbnn_chain_job& J = jobs[q];
        if (!J.ctx || !J.cfg || !J.W_inout || !J.result || !J.out_accepted) return fail(J.ctx, NPBNN_E_ARG, "chains_run_batched: job %d incomplete", q);
        if (!J.ctx->arch_set) return fail(J.ctx, NPBNN_E_STATE, "chains_run_batched: job %d: call npbnn_set_arch first", q);
        for (int p2 = 0; p2 < q; ++p2)
            if (jobs[p2].ctx == J.ctx) return fail(J.ctx, NPBNN_E_ARG, "chains_run_batched: jobs %d and %d share a ctx", p2, q);
        // replicas of one model over the same resident matrix: same device, same X, same network shape, same likelihood
        const npbnn_ctx* a = ctx0;
        const npbnn_ctx* b = J.ctx;
        if (b->device != a->device || b->ds[0].X != a->ds[0].X || b->ds[0].n_rows != a->ds[0].n_rows || b->n_weights != a->n_weights ||
            memcmp(&b->arch, &a->arch, sizeof(npbnn_arch)) != 0 || b->l0_blocks != a->l0_blocks || (b->n_classw > 0) != (a->n_classw > 0) ||
            (b->ds[0].inst_w != nullptr) != (a->ds[0].inst_w != nullptr))
            return fail(J.ctx, NPBNN_E_ARG, "chains_run_batched: job %d is not a replica of job 0 (same device, shared feature matrix "
                                            "(npbnn_share_data), same architecture and likelihood)", q);
        force_f32 |= J.cfg->force_f32;
    }
    HIP_TRY(ctx0, hipSetDevice(ctx0->device));
    Dataset& d0 = ctx0->ds[0];
    int rc = check_dataset_for_lik(ctx0, d0, ctx0->net.lik_kind);
    if (rc) return rc;
    // the group's launch: one candidate per chain, the evaluating workgroups share the tiles, one step workgroup per chain
    LaunchPlan lpG;
    rc = plan_launch(ctx0, 0, &lpG, force_f32, n_jobs, false, true);
    if (rc) return rc;
    if (lpG.n_cand != n_jobs)
        return fail(ctx0, NPBNN_E_ARG, "chains_run_batched: %d weight images do not fit a compute unit's LDS together (%d do)", n_jobs, lpG.n_cand);
    int G = lpG.grid;
    if (G > ctx0->n_cu - n_jobs) G = ctx0->n_cu - n_jobs;
    if (G < 1) G = 1;
    std::vector<ChainBatch> B(n_jobs);
    std::vector<npbnn_chain_cfg> cfgs(n_jobs);
    for (int q = 0; q < n_jobs; ++q) {
        const npbnn_chain_job& J = jobs[q];
        cfgs[q] = *J.cfg;
        cfgs[q].force_f32 = force_f32;
        rc = chain_prepare(J.ctx, &cfgs[q], J.W_inout, J.mask_packed, K, J.M, J.idx, J.delta, J.cnt, J.log_u, 0, &B[q], false, G);
        if (rc) {
            if (J.ctx != ctx0) ctx0->err = J.ctx->err;
            for (int p2 = 0; p2 <= q; ++p2) (void)hipStreamSynchronize(jobs[p2].ctx->stream);
            return rc;
        }
        if ((J.ctx->net.l0_f16 != 0) != (ctx0->net.l0_f16 != 0)) {
            for (int p2 = 0; p2 <= q; ++p2) (void)hipStreamSynchronize(jobs[p2].ctx->stream);
            return fail(ctx0, NPBNN_E_STATE, "chains_run_batched: the chains ended up on different layer-0 paths");
        }
    }
    hipStream_t st = ctx0->stream;
    for (int q = 1; q < n_jobs; ++q) {          // the other chains' preparation (their own streams) before the first group launch
        npbnn_ctx* c = jobs[q].ctx;
        if (!c->ev_x) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_x, c->stream));
        HIP_TRY(ctx0, hipStreamWaitEvent(st, c->ev_x, 0));
    }
    if (!ctx0->d_gparams) {
        HIP_TRY(ctx0, hipMalloc(&ctx0->d_gparams, sizeof(EvalParams)));
        HIP_TRY(ctx0, hipHostMalloc(&ctx0->h_gparams, sizeof(EvalParams)));
    }
    {
        EvalParams g = make_params(ctx0, d0);
        g.partials = nullptr;
        g.inst_w = d0.inst_w;
        g.use_classw = ctx0->n_classw > 0 ? 1 : 0;
        g.has_pass = 0;
        g.group_n = n_jobs;
        for (int q = 0; q < n_jobs; ++q) {
            npbnn_ctx* c = jobs[q].ctx;
            GroupSlot& S = g.group[q];
            S.chain = c->d_cparams;
            S.pass = reinterpret_cast<const PassDesc*>(reinterpret_cast<const char*>(c->d_eparams) + offsetof(EvalParams, pass_desc));
            S.image = c->d_image;
            S.pv = c->d_pv;
            S.pos = c->d_pos;
            S.pscale = c->net.l0_f16 ? c->d_pscale : nullptr;
            S.partials = c->d_partials;
            S.M = jobs[q].M;
        }
        memcpy(ctx0->h_gparams, &g, sizeof g);
        HIP_TRY(ctx0, hipMemcpyAsync(ctx0->d_gparams, ctx0->h_gparams, sizeof g, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(ctx0, hipFuncSetAttribute(reinterpret_cast<const void*>(lpG.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lpG.lds));
    // a launch decides at most one iteration per chain and loses one to every accept (the pass in flight is void): what the chains'
    // last acceptance rates say is needed, then look and repeat if some chain is short
    int launch = 0, rounds = 0;
    std::vector<int> t_done(n_jobs, 0);
    for (;;) {
        int rem = 0;
        double acc = 0.0;
        for (int q = 0; q < n_jobs; ++q) {
            if (K - t_done[q] > rem) rem = K - t_done[q];
            const double a = jobs[q].ctx->accept_rate < 0 ? 0.3 : jobs[q].ctx->accept_rate;
            if (a > acc) acc = a;
        }
        if (rem == 0) break;
        if (++rounds > 64) return fail(ctx0, NPBNN_E_STATE, "chains_run_batched: the chains are stuck");
        const int n = (int)std::ceil((double)rem * (1.0 + acc) * 1.05) + 3;
        for (int i = 0; i < n; ++i, ++launch)
            hipLaunchKernelGGL(lpG.fn, dim3(G + n_jobs), dim3(lpG.wpb * 64), lpG.lds, st, (const EvalParams*)ctx0->d_gparams, launch, 1);
        HIP_TRY(ctx0, hipGetLastError());
        for (int q = 0; q < n_jobs; ++q) {
            npbnn_ctx* c = jobs[q].ctx;
            HIP_TRY(c, hipMemcpyAsync(c->h_res, c->d_res, B[q].RL.total, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(ctx0, hipStreamSynchronize(st));
        for (int q = 0; q < n_jobs; ++q) {
            const ChainDev* now = reinterpret_cast<const ChainDev*>(jobs[q].ctx->h_res);
            if (now->t < t_done[q]) return fail(jobs[q].ctx, NPBNN_E_STATE, "chains_run_batched: chain %d went backwards (t=%d)", q, now->t);
            t_done[q] = now->t;
        }
    }
    for (int q = 0; q < n_jobs; ++q)            // (nothing is handed back unless every chain's batch is good: the caller repeats the
        if (jobs[q].ctx->net.l0_f16 && (*reinterpret_cast<const int*>(jobs[q].ctx->h_res + 448) & kFlagF16Range))      // whole group)
            return fail(ctx0, NPBNN_E_RANGE, "chains_run_batched: a layer-0 weight of chain %d left the fp16 range during this batch", q);
    for (int q = 0; q < n_jobs; ++q) {
        const npbnn_chain_job& J = jobs[q];
        rc = chain_finish(J.ctx, B[q], &cfgs[q], J.W_inout, J.out_accepted, J.out_loglik_prop, J.out_logprior_prop, J.result, K);
        if (rc) {
            if (J.ctx != ctx0) ctx0->err = J.ctx->err;
            return rc;
        }
    }
    return NPBNN_OK;
}

int npbnn_chains_run_exchange(npbnn_comm* comm, npbnn_chain_job* jobs, int32_t n_jobs, int32_t n_chains, int32_t seg_len, int32_t n_seg,
                              const int32_t* swap_j, const int32_t* swap_k, const double* swap_logu, double launch_slack,
                              double* out_records, int32_t* out_segments_done) {
    if (!jobs || n_jobs < 1 || n_jobs > 64 || seg_len < 1 || n_seg < 1 || !swap_j || !swap_k || !swap_logu || !out_segments_done)
        return fail(nullptr, NPBNN_E_ARG, "chains_run_exchange: bad arguments");
    if ((long long)seg_len * n_seg > (1 << 24)) return fail(nullptr, NPBNN_E_ARG, "chains_run_exchange: %d x %d iterations in one call", n_seg, seg_len);
    int device = -1, rank = 0, world = 1;
    if (comm) {
        int rc = npbnn_comm_info_(comm, &device, &rank, &world);
        if (rc) return rc;
    }
    if (n_chains != world * n_jobs)
        return fail(nullptr, NPBNN_E_ARG, "chains_run_exchange: %d chains on %d ranks x %d jobs (every rank must hold the same number)", n_chains, world, n_jobs);
    for (int q = 0; q < n_jobs; ++q) {
        const npbnn_chain_job& J = jobs[q];
        if (!J.ctx || !J.cfg || !J.W_inout || !J.result || !J.out_accepted) return fail(J.ctx, NPBNN_E_ARG, "chains_run_exchange: job %d incomplete", q);
        if (device < 0) device = J.ctx->device;
        if (J.ctx->device != device) return fail(J.ctx, NPBNN_E_ARG, "chains_run_exchange: job %d is on device %d, the others on %d", q, J.ctx->device, device);
        if (J.chain_id != rank + world * q) return fail(J.ctx, NPBNN_E_ARG, "chains_run_exchange: job %d holds chain %d, expected %d", q, J.chain_id, rank + world * q);
        for (int p2 = 0; p2 < q; ++p2)
            if (jobs[p2].ctx == J.ctx) return fail(J.ctx, NPBNN_E_ARG, "chains_run_exchange: jobs %d and %d share a ctx", p2, q);
    }
    for (int s = 0; s < n_seg; ++s)
        if (swap_j[s] < 0 || swap_j[s] >= n_chains || swap_k[s] < 0 || swap_k[s] >= n_chains)
            return fail(jobs[0].ctx, NPBNN_E_ARG, "chains_run_exchange: swap %d names chains %d, %d of %d", s, swap_j[s], swap_k[s], n_chains);
    if (!(launch_slack > 0.0)) launch_slack = 1.25;       // (< 1 starves the segments on purpose: tests of the shortfall path)
    const int K = seg_len * n_seg;
    npbnn_ctx* ctx0 = jobs[0].ctx;
    HIP_TRY(ctx0, hipSetDevice(device));
    const auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
    // per-ctx exchange block; the records live in job 0's and are shared
    struct XLayout { size_t sj, sk, su, state, rec, cold, total; };
    std::vector<XLayout> XL(n_jobs);
    std::vector<ChainBatch> B(n_jobs);
    const size_t rec_bytes = (size_t)n_seg * n_chains * kRecDoubles * sizeof(double);
    int rc_prepare = NPBNN_OK;       // first failure of this rank before anything is enqueued (the ranks agree on it below)
    auto size_exchange_block = [&](int q) -> int {
        npbnn_ctx* ctx = jobs[q].ctx;
        XLayout& L = XL[q];
        L.sj = 256;
        L.sk = L.sj + up256((size_t)n_seg * sizeof(int));
        L.su = L.sk + up256((size_t)n_seg * sizeof(int));
        L.state = L.su + up256((size_t)n_seg * sizeof(double));
        L.rec = L.state + up256((size_t)n_seg * 4 * sizeof(double));
        L.cold = L.rec + (q == 0 ? up256(rec_bytes) : 0);
        L.total = L.cold + (jobs[q].out_cold_w ? up256((size_t)n_seg * ctx->n_weights * sizeof(double)) : 0);
        static_assert(sizeof(ExchangeParams) <= 256, "ExchangeParams must fit its slot");
        if (L.total > ctx->xbuf_cap) {
            if (ctx->d_xbuf) (void)hipFree(ctx->d_xbuf);
            if (ctx->h_xbuf) (void)hipHostFree(ctx->h_xbuf);
            ctx->d_xbuf = nullptr; ctx->h_xbuf = nullptr; ctx->xbuf_cap = 0;
            const size_t cap = L.total + L.total / 2;
            HIP_TRY(ctx, hipMalloc(&ctx->d_xbuf, cap));
            HIP_TRY(ctx, hipHostMalloc(&ctx->h_xbuf, cap));
            ctx->xbuf_cap = cap;
        }
        if (!ctx->ev_x) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming));
        return NPBNN_OK;
    };
    for (int q = 0; q < n_jobs && !rc_prepare; ++q) {
        rc_prepare = size_exchange_block(q);
        if (rc_prepare && jobs[q].ctx != ctx0) ctx0->err = jobs[q].ctx->err;
    }
    double* d_rec = rc_prepare ? nullptr : reinterpret_cast<double*>(ctx0->d_xbuf + XL[0].rec);
    // Everything that can fail on this rank alone (bad draws, a weight outside the fp16 range, allocation) is checked BEFORE the
    // first collective is enqueued, and the ranks agree on the outcome (one host-side all-gather): either every rank enqueues
    // its in-place all-gathers, or none does.
    for (int q = 0; q < n_jobs && !rc_prepare; ++q) {
        const npbnn_chain_job& J = jobs[q];
        npbnn_ctx* ctx = J.ctx;
        int rc = chain_prepare(ctx, J.cfg, J.W_inout, J.mask_packed, K, J.M, J.idx, J.delta, J.cnt, J.log_u, seg_len, &B[q], n_jobs == 1);
        if (rc) {
            if (ctx != ctx0) ctx0->err = ctx->err;
            for (int p2 = 0; p2 <= q; ++p2) (void)hipStreamSynchronize(jobs[p2].ctx->stream);
            rc_prepare = rc;
            break;
        }
        const XLayout& L = XL[q];
        ExchangeParams x{};
        x.rec = d_rec;
        x.swap_j = reinterpret_cast<const int*>(ctx->d_xbuf + L.sj);
        x.swap_k = reinterpret_cast<const int*>(ctx->d_xbuf + L.sk);
        x.swap_logu = reinterpret_cast<const double*>(ctx->d_xbuf + L.su);
        x.snap_state = reinterpret_cast<double*>(ctx->d_xbuf + L.state);
        x.snap_w = J.out_cold_w ? reinterpret_cast<double*>(ctx->d_xbuf + L.cold) : nullptr;
        x.world = world;
        x.per_rank = n_jobs;
        x.n_seg = n_seg;
        x.seg_len = seg_len;
        x.my_slot = rank * n_jobs + q;
        x.n_weights = ctx->n_weights;
        memset(ctx->h_xbuf, 0, 256);
        memcpy(ctx->h_xbuf, &x, sizeof x);
        memcpy(ctx->h_xbuf + L.sj, swap_j, (size_t)n_seg * sizeof(int));
        memcpy(ctx->h_xbuf + L.sk, swap_k, (size_t)n_seg * sizeof(int));
        memcpy(ctx->h_xbuf + L.su, swap_logu, (size_t)n_seg * sizeof(double));
        if (hipMemcpyAsync(ctx->d_xbuf, ctx->h_xbuf, L.state, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            (x.snap_w && hipMemsetAsync(ctx->d_xbuf + L.cold, 0, (size_t)n_seg * ctx->n_weights * sizeof(double), ctx->stream) != hipSuccess))
            rc_prepare = fail(ctx0, NPBNN_E_HIP, "chains_run_exchange: staging the exchange block of job %d failed", q);
    }
    if (comm && world > 1) {
        std::vector<double> ok_all((size_t)world, 0.0);
        const double ok_mine = rc_prepare == NPBNN_OK ? 1.0 : 0.0;
        int rc = npbnn_comm_allgather_f64(comm, &ok_mine, 1, ok_all.data());
        if (rc) {
            ctx0->err = npbnn_last_error(nullptr);
            for (int q = 0; q < n_jobs; ++q) (void)hipStreamSynchronize(jobs[q].ctx->stream);
            return rc;
        }
        int bad_rank = -1;
        for (int r = 0; r < world; ++r)
            if (ok_all[(size_t)r] != 1.0 && bad_rank < 0) bad_rank = r;
        if (bad_rank >= 0) {
            for (int q = 0; q < n_jobs; ++q) (void)hipStreamSynchronize(jobs[q].ctx->stream);
            if (rc_prepare) return rc_prepare;
            return fail(ctx0, NPBNN_E_COMM, "chains_run_exchange: rank %d could not prepare its chains; nothing was enqueued on any rank", bad_rank);
        }
    } else if (rc_prepare) {
        return rc_prepare;
    }
    HIP_TRY(ctx0, hipMemsetAsync(d_rec, 0, rec_bytes, ctx0->stream));
    if (n_jobs > 1) {      // the other chains' first record must not overtake the clearing of the shared block
        HIP_TRY(ctx0, hipEventRecord(ctx0->ev_x, ctx0->stream));
        for (int q = 1; q < n_jobs; ++q) HIP_TRY(jobs[q].ctx, hipStreamWaitEvent(jobs[q].ctx->stream, ctx0->ev_x, 0));
    }
    const int rec_per_rank = n_jobs * kRecDoubles;
    std::vector<hipStream_t> xs(n_jobs);        // stream the exchange kernels of a job run on
    for (int s = 0; s < n_seg; ++s) {
        for (int q = 0; q < n_jobs; ++q) {
            npbnn_ctx* ctx = jobs[q].ctx;
            int rc = chain_enqueue(ctx, B[q], passes_for_segment(ctx, B[q], seg_len, launch_slack));
            if (rc) {       // the peers have this interval's collective in flight: it must not pair with anything else we issue
                if (ctx != ctx0) ctx0->err = ctx->err;
                if (comm && world > 1) npbnn_comm_abort_(comm);
                (void)hipDeviceSynchronize();
                return rc;
            }
            // two-stream schedule (one chain on this GPU): the exchange kernels follow the interval's LAST launch on its stream -
            // by then the other stream's launches are through as well (the last step waited for them) - and the next interval's
            // first launch, which goes to the other stream, waits behind a gate for exchange_apply_kernel's hand-over.  No host
            // synchronisation, no stream events.
            xs[q] = B[q].sync ? ctx->stream_e[(B[q].launch - 1) & 1] : ctx->stream;
            hipLaunchKernelGGL(exchange_pack_kernel, dim3(1), dim3(64), 0, xs[q], (const ChainParams*)ctx->d_cparams,
                               (const ExchangeParams*)ctx->d_xbuf, s);
            if (q > 0) {
                HIP_TRY(ctx, hipEventRecord(ctx->ev_x, ctx->stream));
                HIP_TRY(ctx0, hipStreamWaitEvent(ctx0->stream, ctx->ev_x, 0));
            }
        }
        if (comm) {
            int rc = npbnn_comm_allgather_inplace_stream_(comm, d_rec + (size_t)s * n_chains * kRecDoubles, rec_per_rank, xs[0]);
            if (rc) {
                ctx0->err = npbnn_last_error(nullptr);
                if (world > 1) npbnn_comm_abort_(comm);
                for (int q = 0; q < n_jobs; ++q) (void)hipStreamSynchronize(jobs[q].ctx->stream);
                return rc;
            }
        }
        if (n_jobs > 1) {
            HIP_TRY(ctx0, hipEventRecord(ctx0->ev_x, ctx0->stream));
            for (int q = 1; q < n_jobs; ++q) HIP_TRY(jobs[q].ctx, hipStreamWaitEvent(jobs[q].ctx->stream, ctx0->ev_x, 0));
        }
        for (int q = 0; q < n_jobs; ++q) {
            npbnn_ctx* ctx = jobs[q].ctx;
            hipLaunchKernelGGL(exchange_apply_kernel, dim3(1), dim3(1024), 0, xs[q], (const ChainParams*)ctx->d_cparams,
                               (const ExchangeParams*)ctx->d_xbuf, s, B[q].launch, B[q].overlap ? 1 : 0);
            if (B[q].sync && s + 1 < n_seg)
                hipLaunchKernelGGL(sync_gate_exchanged_kernel, dim3(1), dim3(64), 0, ctx->stream_e[B[q].launch & 1], ctx->d_chain, s + 1);
        }
    }
    for (int q = 0; q < n_jobs; ++q) {
        int rc = chain_join(jobs[q].ctx, B[q]);
        if (rc) return rc;
    }
    for (int q = 0; q < n_jobs; ++q) {
        npbnn_ctx* ctx = jobs[q].ctx;
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_res, ctx->d_res, B[q].RL.total, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_xbuf + XL[q].state, ctx->d_xbuf + XL[q].state, XL[q].total - XL[q].state, hipMemcpyDeviceToHost, ctx->stream));
    }
    for (int q = 0; q < n_jobs; ++q) {
        if (comm && world > 1) {       // these streams wait for collectives: a peer that has gone must not hang this rank (wait_stream)
            int rc = npbnn_comm_wait_stream_(comm, jobs[q].ctx->stream, "chains_run_exchange");
            if (rc) {
                ctx0->err = npbnn_last_error(nullptr);
                (void)hipDeviceSynchronize();           // (the communicator was aborted: its kernels leave, the streams drain)
                return rc;
            }
        } else {
            HIP_TRY(jobs[q].ctx, hipStreamSynchronize(jobs[q].ctx->stream));
        }
    }
    int seg_done = -1;
    for (int q = 0; q < n_jobs; ++q) {
        const ChainDev* fin = reinterpret_cast<const ChainDev*>(jobs[q].ctx->h_res);
        if (fin->aborted) jobs[q].ctx->sync_failed = true;     // (the chain stopped at a valid state; the records show it as short)
        if (seg_done < 0) seg_done = fin->seg_idx;
        if (fin->seg_idx != seg_done) return fail(ctx0, NPBNN_E_STATE, "chains_run_exchange: chains disagree on the exchanges done (%d, %d)", seg_done, fin->seg_idx);
        if (fin->t < seg_done * seg_len || fin->t > K) return fail(ctx0, NPBNN_E_STATE, "chains_run_exchange: chain %d is at iteration %d after %d exchanges", q, fin->t, seg_done);
    }
    for (int q = 0; q < n_jobs; ++q) {
        const npbnn_chain_job& J = jobs[q];
        npbnn_ctx* ctx = J.ctx;
        const ChainDev* fin = reinterpret_cast<const ChainDev*>(ctx->h_res);
        int rc = chain_finish(ctx, B[q], J.cfg, J.W_inout, J.out_accepted, J.out_loglik_prop, J.out_logprior_prop, J.result, fin->t, true);
        if (rc) return rc;
        if (J.out_state) memcpy(J.out_state, ctx->h_xbuf + XL[q].state, (size_t)n_seg * 4 * sizeof(double));
        if (J.out_cold_w) memcpy(J.out_cold_w, ctx->h_xbuf + XL[q].cold, (size_t)n_seg * ctx->n_weights * sizeof(double));
    }
    if (out_records) {     // device order (rank-major) -> chain order
        const double* h_rec = reinterpret_cast<const double*>(ctx0->h_xbuf + XL[0].rec);
        for (int s = 0; s < n_seg; ++s)
            for (int i = 0; i < n_chains; ++i) {
                const int slot = (i % world) * n_jobs + i / world;
                memcpy(out_records + ((size_t)s * n_chains + i) * kRecDoubles, h_rec + ((size_t)s * n_chains + slot) * kRecDoubles,
                       kRecDoubles * sizeof(double));
            }
    }
    *out_segments_done = seg_done;
    return NPBNN_OK;
}

// diagnostics (not part of the ABI): in the next batch of this context that runs on the two-stream schedule, the step workgroup of
// launch `launch` never reports back - every wait behind it must time out cleanly (tests of that path)
int npbnn_debug_sync_skip_(npbnn_ctx* ctx, int launch) {
    if (!ctx) return NPBNN_E_ARG;
    ctx->debug_sync_skip = launch;
    return NPBNN_OK;
}

// diagnostics (not part of the ABI): the weight image as it stands in device memory
int npbnn_debug_image_(npbnn_ctx* ctx, float* out, int n) {
    if (!ctx || !out) return NPBNN_E_ARG;
    if (n > ctx->net.image_floats) n = ctx->net.image_floats;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_image, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
}

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
        HIP_TRY(ctx, hipMalloc(&d_stamps, (size_t)lp.grid * 24 * sizeof(unsigned long long)));     // [grid][8] wave 0 + [grid][16] per wave
        HIP_TRY(ctx, hipMemset(d_stamps, 0, (size_t)lp.grid * 24 * sizeof(unsigned long long)));
        p.stamps = d_stamps;
    }
    rc = push_eval_params(ctx, p);
    if (rc) return rc;
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
    if (d_stamps) {
        std::vector<unsigned long long> hs((size_t)lp.grid * 24);
        (void)hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d_stamps);
        {   // when each wave of a workgroup finished its tiles, relative to the workgroup's tile-loop start (mean over workgroups)
            double done[16] = {0};
            for (int b = 0; b < lp.grid; ++b)
                for (int w = 0; w < lp.wpb && w < 16; ++w)
                    done[w] += (double)(hs[(size_t)lp.grid * 8 + (size_t)b * 16 + w] - hs[(size_t)b * 8 + 3]) * 0.01;
            fprintf(stderr, "[npbnn eval stamps] tiles done per wave, us after the tile loop starts:");
            for (int w = 0; w < lp.wpb && w < 16; ++w) fprintf(stderr, " %.1f", done[w] / lp.grid);
            fprintf(stderr, "\n");
        }
        unsigned long long first = ~0ull, last = 0, first_end = ~0ull;
        double acc[8] = {0};
        for (int b = 0; b < lp.grid; ++b) {
            const unsigned long long* q = &hs[(size_t)b * 8];
            if (q[0] < first) first = q[0];
            if (q[6] > last) last = q[6];
            if (q[6] < first_end) first_end = q[6];
            for (int k = 1; k <= 6; ++k) acc[k] += (double)(q[k] - q[k - 1]) * 0.01;      // 100 MHz wall clock -> us
            acc[7] += (double)q[7] * 0.01;
        }
        double late = 0;
        for (int b = 0; b < lp.grid; ++b) late += (double)(hs[(size_t)b * 8] - first) * 0.01;
        fprintf(stderr, "[npbnn eval stamps] wave 0 of a workgroup, mean us: start skew %.2f | issue %.2f  barrier1 %.2f  patch %.2f  tiles %.2f  "
                        "barrier2 %.2f  partials %.2f | tails within tiles %.2f | first start -> first end %.2f, -> last end %.2f\n",
                late / lp.grid, acc[1] / lp.grid, acc[2] / lp.grid, acc[3] / lp.grid, acc[4] / lp.grid, acc[5] / lp.grid, acc[6] / lp.grid, acc[7] / lp.grid,
                (double)(first_end - first) * 0.01, (double)(last - first) * 0.01);
    }
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
    rc = plan_launch(ctx, 0, &lp, 0, 1, false, true);
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
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float burst = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&burst, ctx->ev[0], ctx->ev[1]));
    const double sum = (double)burst;
    // (2) evaluation = eval kernel + finalize
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, ctx->stream, (const EvalParams*)ctx->d_eparams, 0, 1);
        hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, ctx->stream, (const FinalizeParams*)ctx->d_fparams);
    }
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
