// Host-side state of a chain context and the helpers the translation units of the C ABI share
// (npbnn_capi.hip: contexts, data, architecture, evaluation, prediction, timing hooks; npbnn_chain_api.hip: device-resident
// chains, group passes, exchange runs; npbnn_general.hip: the general device chain).  Not part of the ABI.
#pragma once
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
eval_fn_t pick_eval_d1_cat_plain(int mt0, int f16);      // (plain evaluations: no chain pass)
eval_fn_t pick_eval_d1_cat_spec(int mt0, int f16);       // (NPBNN_SCHED_PERSIST_SERIAL: with the outcome-speculative step; nullptr where none)
eval_fn_t pick_eval_d1_gauss_spec(int mt0, int f16);
eval_fn_t pick_eval_d3_cat_spec(int mt0, int f16);
eval_fn_t pick_eval_d3_gauss_spec(int mt0, int f16);
eval_fn_t pick_eval_d1_gauss_plain(int mt0, int f16);
eval_fn_t pick_eval_d1_gauss_fast(int mt0, int f16);
eval_fn_t pick_eval_d2_cat_fast(int mt0, int f16);
eval_fn_t pick_eval_d2_gauss_fast(int mt0, int f16);
eval_fn_t pick_eval_d3_cat_fast(int mt0, int f16);
eval_fn_t pick_eval_d3_gauss_fast(int mt0, int f16);
}

// lk: likelihood class of the build (npbnn::lik_class); the float64 row-wise class has single-candidate builds only
static inline eval_fn_t npbnn_pick_eval_kernel(int mt0, int mti, int f16, int n_cand, int lk, bool fast = false, bool blocked = false, bool plain = false) {
    using namespace npbnn;
    if (fast) {
        const bool g = lk == kLikGauss;
        if (blocked) f16 = 2;                 // (fast_launch_ok: block structure only on the fp16-split path)
        if (n_cand <= 1 && plain) return g ? pick_eval_d1_gauss_plain(mt0, f16) : pick_eval_d1_cat_plain(mt0, f16);
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

namespace npbnn_api {

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
    float* X16w = nullptr;     // the fp16-split copy in the weight-streamed path's piece order (split_x_tiled_kernel), built when a
    bool x16w_borrowed = false;   // network on that path first asks for it; a borrower of X uses (and, if need be, builds) its owner's
};

}  // namespace npbnn_api
using npbnn_api::Dataset;

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
    const void* attr_fn = nullptr; // kernel whose dynamic-LDS limit was raised last, and to what
    size_t attr_lds = 0;
    // layer-0 block structure (npbnn_set_layer_mask): which (16-node tile, 16-feature group) blocks of the mask hold a nonzero;
    // empty = dense
    std::vector<unsigned char> l0_blocks;      // [mt][ceil(in_dim / 16)]
    float* d_xscale = nullptr;     // per-feature power-of-two scales of the fp16-split path (from the training matrix)
    float* d_wscale = nullptr;
    int f16_shifted_cols = 0;      // columns whose scale was moved up (heavy tails: ensure_scales) and the largest such move (powers of two)
    int f16_max_shift = 0;
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
    int last_schedule = 0;         // schedule of the previous batch
    int turn_batches[2][2] = {{0, 0}, {0, 0}};   // [form][batch-size class] batches run since that form was last measured in that class (kTurnReprobeBatches)
    double turn_us[2] = {0.0, 0.0}; // measured time of a launch turn (pass, decided or void) of the persistent forms: overlapped, decision between passes
    int it_n[2][2] = {{0, 0}, {0, 0}};           // [form][batch-size class] batches that went into it_us (the first one of a form is slow: a second follows it)
    double it_us[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // [form][batch-size class] measured time per ITERATION of a batch (what NPBNN_SCHED_AUTO
                                    // compares); classes: batches of fewer than kShortBatch iterations, and the others - a form's fixed
                                    // cost per batch differs, so what wins in dispatches of 100 need not win in sub-batches of 512
    double its_per_pass = 0.0;     // iterations a launch decided on average in the previous batch (0: unknown)
    double accept_rate = -1.0;     // acceptance rate of the previous batch (< 0: unknown)
    double* d_wcur = nullptr;
    double* d_pv = nullptr;        // [kMaxCand][M] proposed values of the candidates in flight
    size_t pv_cap = 0;
    // NPBNN_SCHED_PERSIST_SERIAL (spec_round): outcome-speculative preparation
    SpecState* d_spec = nullptr;
    double* d_spec_pv = nullptr;   // [3][kSpecOutcomes][kMaxCand][M]
    size_t spec_pv_cap = 0;        // M capacity
    unsigned* d_spec_touch = nullptr;   // [kMaxCand][n_weights][4] touch tables: pass tag (cleared before it could repeat), -, value
    size_t spec_touch_cap = 0;     // weights capacity
    unsigned long long* d_spec_part = nullptr;   // [2][kMaxCand][kPartialStride][256][2] the workgroups' sums of a pass as tagged word pairs (npbnn_chain.hip.h, SpecPart)
    unsigned spec_gen = 0;         // pass tags handed out so far
    // rows split over the ranks of a communicator (npbnn_set_row_shard): the records of partial sums are gathered before every step
    int shard_n = 0, shard_rank = 0;
    long long shard_rows_total = 0;
    npbnn_comm* shard_comm = nullptr;
    npbnn_gather_fn shard_gather = nullptr;
    void* shard_user = nullptr;
    double* d_shard_recv = nullptr;   // [shard_n][kMaxCand][kPartialStride]: this rank's record at its own place, the peers' after the gather
    double* d_shard_part = nullptr;   // [kMaxCand][kPartialStride][shard_n]: the same in the layout the step kernel sums
    double* h_shard = nullptr;        // pinned, as d_shard_recv (host-staged gather)
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
    // weight-streamed path (npbnn_wide.hip): the network does not fit a compute unit's LDS (or NPBNN_OPT_WIDE asks for it)
    bool wide = false;
    int wide_option = 0;           // NPBNN_OPT_WIDE: 0 when the resident path cannot hold the network, 1 always
    WideMeta wmeta{};
    float* d_wide_cand = nullptr;  // candidate image of a device chain (the committed image with the pending proposal patched in)
    float* d_wide_act[3] = {nullptr, nullptr, nullptr};   // hidden activations [rows][16 * tiles], ping-pong between layers; [2]: the K-slices' sums
    size_t wide_act_cap = 0;       // floats each
    double* d_prep_terms = nullptr;       // [kMaxCand][M] prior terms of the pending candidates' entries (ChainParams::prep_terms)
    size_t prep_cap = 0;
    WideCandState* d_wide_cs = nullptr;   // what the candidate image's last patch covered (wide proposals: wide_cand_sync)
    unsigned conf_cap = 0;         // classes d_conf / h_conf are sized for
};

static_assert(sizeof(EvalParams) % 8 == 0 && sizeof(FinalizeParams) % 8 == 0, "parameter blocks are laid out back to back");

namespace npbnn_api {

// NPBNN_SCHED_AUTO between the two persistent forms.  Overlapped (PERSIST): a pass that accepts voids the pass in flight behind it - a
// decided pass costs (1 + a) turns, a = 1 - (1 - p)^D the share of passes that accept something.  Decision between the passes
// (PERSIST_SERIAL): no pass in vain, but every turn is longer by the decision and by what the step workgroup cannot hide of preparing
// the next pass for every outcome - more, the wider the proposals.  The context keeps what an iteration cost on each form
// (npbnn_ctx.it_us: a batch's time over its iterations) and picks the cheaper one; a form that has not run yet is priced from the other
// one with that model and kSpecTurnExtraUs + kSpecTurnExtraUsPerWeight * M (measured on config-2 shapes: 33.2 against 28.5 us per turn
// at M = 33, 40 at M = 428), and is given its first batches after kTurnFirstProbeBatches batches on the other when the model puts it within kFirstProbeWithin of it, or after kTurnForcedProbeBatches regardless (then
// one every kTurnReprobeBatches while within a factor two: measurements go stale).  Short batches (dispatches of 100) and long ones (the
// sub-batches of a long call) are measured and decided apart.
constexpr double kSpecTurnExtraUs = 4.5, kSpecTurnExtraUsPerWeight = 0.0175;
constexpr double kTurnUsGuess = 30.0;           // before anything has been measured
constexpr int kShortBatch = 256;                // batches below / from this many iterations are measured (and decided) apart
constexpr double kFirstProbeWithin = 1.15;      // ... if the model prices it within this factor of the measured one (asked again at every batch: the acceptance rate moves)
constexpr int kTurnForcedProbeBatches = 200;    // ... or whatever the model says after this many batches without it (a wrong model must not park a chain for good)
constexpr int kTurnFirstProbeBatches = 4;       // batches on one persistent form before the other, never measured, is given one
constexpr int kTurnReprobeBatches = 48;         // batches on one persistent form before the other's measured turn time is refreshed
constexpr int kPersistSerialMaxWidth = 640;     // ... and the widest proposal (weights perturbed per iteration) it is picked for
constexpr int kWideMaxCand = 3;                // weight sets a fused pass of the weight-streamed path carries at most
constexpr int kMinResidentWaves = 4;           // fewer waves than this beside the weight image: the network runs on the weight-streamed path
constexpr int kWideStepPatchMax = 2048;      // widest proposal whose candidate image the step workgroup keeps by itself (weight-streamed path)
constexpr size_t kChainMinCapacity = 2048;    // iterations the per-batch chain buffers are sized for at least (allocation is slow)

int fail(npbnn_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, NPBNN_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct LaunchPlan {
    eval_fn_t fn;
    eval_fn_t fn_spec;       // the same launch's build with the outcome-speculative step (NPBNN_SCHED_PERSIST_SERIAL), or nullptr
    int n_cand;
    int grid, wpb;
    size_t lds;
    int n_waves;
    bool fast;
    bool wide = false;       // weight-streamed path: no single kernel - wide_forward launches the layers' products and the likelihood
};

// ---- weight-streamed path (npbnn_wide.hip) ----
bool wide_needed(const npbnn_ctx* ctx, const npbnn_arch* a, bool f16);
int wide_build(npbnn_ctx* ctx, bool f16);
void wide_free(npbnn_ctx* ctx);
int wide_plan(npbnn_ctx* ctx, int which, LaunchPlan* lp, int want_cand = 1);
void wide_pack(npbnn_ctx* ctx, const double* d_w, const double* d_col_override, float* image, int* flags);
// the forward pass + likelihood of the weights in `image` on the ctx stream; the launch's EvalParams must be in ctx->d_eparams.
// chain_pass: a pass of a device chain (the kernels leave at once when the chain's batch is through; candidate slopes from the chain)
// only_layer0: stop behind the first layer's product (timing hook); info: that product's geometry, or nullptr
int wide_forward(npbnn_ctx* ctx, int which, const float* image, bool chain_pass, bool only_layer0 = false, int* info = nullptr, int n_cand = 1);
int wide_cand_begin(npbnn_ctx* ctx);     // start of a chain batch: candidate image = committed image, nothing patched
void wide_cand_sync(npbnn_ctx* ctx, int M, int n_cand, bool make_them);   // before a pass of a chain with wide proposals: candidate image = committed image + the pending proposal
int ensure_conf(npbnn_ctx* ctx, int n_classes);
// one evaluation launch of a plan on the ctx stream (resident: the plan's kernel; weight-streamed: wide_forward on the committed image)
int launch_plain_eval(npbnn_ctx* ctx, const LaunchPlan& lp, int which);

int max_inner_tiles(const NetMeta& net);
WaveLayout layout_for(const npbnn_ctx* ctx, const Dataset& d, bool predict_only = false);
int pick_waves_per_block(const npbnn_ctx* ctx, size_t* lds_bytes, int n_cand, const WaveLayout& lay, bool predict_only = false, bool fast = false);
int ensure_x16(npbnn_ctx* ctx, int which, int* usable);
int rebuild_net(npbnn_ctx* ctx, bool f16);
bool l0_blocked(const NetMeta& net);
bool fast_launch_ok(const npbnn_ctx* ctx, const Dataset& d);
// lik_only: the caller wants the likelihood terms and nothing else from the launch (no statistics, no predictions)
// plain: the launch is a plain evaluation - no pass descriptor, no chain (the builds without that code)
int plan_launch(npbnn_ctx* ctx, int which, LaunchPlan* lp, int force_f32 = 0, int want_cand = 1, bool predict_only = false, bool lik_only = false,
                bool plain = false);
int ensure_work_buffers(npbnn_ctx* ctx, int n_waves);
int stage_weights(npbnn_ctx* ctx, const double* W, const double* act_prm, const double* col_override);
int push_eval_params(npbnn_ctx* ctx, const EvalParams& p);
int push_finalize_params(npbnn_ctx* ctx, const FinalizeParams& f);
int push_chain_params(npbnn_ctx* ctx, const ChainParams& c);
EvalParams make_params(npbnn_ctx* ctx, const Dataset& d);
void report_eval_stamps(unsigned long long* d_stamps, int grid, int wpb, int first_wg);
int check_dataset_for_lik(npbnn_ctx* ctx, const Dataset& d, int lik);
double wall_us();
// launches of kernels that live in npbnn_capi.hip, for the other translation units: the weight image of device-resident float64
// weights (col_override may be nullptr) into `image`, and the reduction of the partial sums of the last evaluation
void launch_pack_weights(npbnn_ctx* ctx, const double* d_w, const double* d_col_override, float* image, int* flags);
void launch_finalize(npbnn_ctx* ctx);

}  // namespace npbnn_api
using namespace npbnn_api;
