// Device-resident chains behind the C ABI (include/npbnn_hip.h): npbnn_chain_run, the group pass, the exchange run.
// Host-side orchestration only: uploads, the batch's launches on the chain's stream(s), the result copy.
#define NPBNN_KERNELS_CHAIN
#include "npbnn_ctx.hip.h"

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
    bool make_cands = false;                   // weight-streamed path: the candidates are made between step and pass (ChainParams::prep_terms)
    size_t wb = 0;
    int launch = 0;                            // launches enqueued so far (overlapped schedule: the pass parity follows it)
    unsigned long long* d_stamps = nullptr;    // diagnostics
    unsigned long long* d_estamps = nullptr;   // diagnostics (NPBNN_EVAL_STAMPS: the evaluating workgroups' phases, last pass of the batch)
    double tw0 = 0.0, tw1 = 0.0;
};

// NPBNN_CHAIN_TIMING=2: synchronise after every stage of a batch's set-up and print what each took (diagnostics: it serialises
// what normally overlaps)
static int stage_timing() { static const int v = getenv("NPBNN_CHAIN_TIMING") ? atoi(getenv("NPBNN_CHAIN_TIMING")) : 0; return v >= 2; }
static void stage_mark(npbnn_ctx* ctx, const char* what, double* t_last) {
    if (!stage_timing()) return;
    (void)hipStreamSynchronize(ctx->stream);
    const double now = wall_us();
    fprintf(stderr, "[npbnn stage] %-28s %.1f us\n", what, now - *t_last);
    *t_last = wall_us();
}

// ---- rows split over ranks (npbnn_set_row_shard) ----
// this rank's record of a pass: per candidate and value in use, the sum of the pass's per-wave partials in the order the step kernel
// itself would add them (lanes take records lane, lane + 64, ... in turn, then a fixed butterfly) - written to the rank's own place in
// the gather buffer [rank][candidate][value]
__global__ void __launch_bounds__(1024) shard_sum_kernel(const double* __restrict__ partials, int n_blocks, int D, int lik_kind, int k_targets,
                                                         double* __restrict__ mine) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nvals = partial_value_count(lik_kind, k_targets);
    for (int item = wave; item < D * nvals; item += nw) {
        const int j = item / nvals, v = partial_value_index(item % nvals, k_targets);
        const double* src = partials + ((size_t)j * kPartialStride + v) * n_blocks;
        double s = 0.0;
        for (int b = lane; b < n_blocks; b += 64) s += src[b];
        s = butterfly_sum_f64(s);
        if (lane == 0) mine[j * kPartialStride + v] = s;
    }
}
// the gathered records [rank][candidate][value] in the layout the step kernel sums: [candidate][value][rank] (rank order = the order
// of the addition, the same on every rank)
__global__ void __launch_bounds__(256) shard_spread_kernel(const double* __restrict__ gathered, int n_ranks, double* __restrict__ part) {
    const int rec = kMaxCand * kPartialStride;
    for (int i = threadIdx.x; i < rec * n_ranks; i += blockDim.x) {
        const int r = i / rec, e = i % rec;
        part[(size_t)e * n_ranks + r] = gathered[i];
    }
}

// between a pass and its step: every rank's record to every rank, on the chain's stream (RCCL) or through the host (gather callback)
int shard_exchange(npbnn_ctx* ctx, const LaunchPlan& lp, int D) {
    hipStream_t st = ctx->stream;
    const int rec = kMaxCand * kPartialStride;
    hipLaunchKernelGGL(shard_sum_kernel, dim3(1), dim3(1024), 0, st, (const double*)ctx->d_partials, lp.n_waves, D, ctx->net.lik_kind, ctx->net.k_targets,
                       ctx->d_shard_recv + (size_t)ctx->shard_rank * rec);
    if (ctx->shard_n > 1) {
        if (ctx->shard_comm) {
            const int rc = npbnn_comm_allgather_inplace_stream_(ctx->shard_comm, ctx->d_shard_recv, rec, st);
            if (rc) return fail(ctx, NPBNN_E_COMM, "chain_run: the gather of the row shards' sums failed");
        } else {
            double* mine = ctx->h_shard + (size_t)ctx->shard_n * rec;          // (send block behind the receive block)
            HIP_TRY(ctx, hipMemcpyAsync(mine, ctx->d_shard_recv + (size_t)ctx->shard_rank * rec, rec * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (ctx->shard_gather(ctx->shard_user, mine, ctx->h_shard, rec) != 0)
                return fail(ctx, NPBNN_E_COMM, "chain_run: the gather callback of the row shards failed");
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_shard_recv, ctx->h_shard, (size_t)ctx->shard_n * rec * sizeof(double), hipMemcpyHostToDevice, st));
        }
    }
    hipLaunchKernelGGL(shard_spread_kernel, dim3(1), dim3(256), 0, st, (const double*)ctx->d_shard_recv, ctx->shard_n, ctx->d_shard_part);
    return NPBNN_OK;
}

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
    // (the K * M weight indices are checked on the device, where they are read anyway: gather_pos_kernel)
    Dataset& d = ctx->ds[0];
    int rc = check_dataset_for_lik(ctx, d, lik);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchPlan& lp = B->lp;
    int want_cand = cfg->n_candidates;
    if (want_cand < 1) want_cand = kMaxCand;            // 0 = as many as fit
    if (group_blocks > 0) want_cand = 1;
    if (cfg->slope_idx && ctx->wide) want_cand = 1;      // (weight-streamed path: candidates with slopes of their own travel one per pass)
    rc = plan_launch(ctx, 0, &lp, cfg->force_f32, want_cand, false, true);
    if (rc) return rc;
    const int D = lp.n_cand;
    const bool sharded = ctx->shard_n > 0;
    if (sharded && (group_blocks > 0 || seg_len > 0))
        return fail(ctx, NPBNN_E_STATE, "chain_run: a context whose rows are split over ranks runs plain batches only (no group pass, no exchange run)");
    // schedule: overlapping the decision of a pass with the evaluation of the next pays as long as most passes reject everything
    if (lp.wide && group_blocks > 0)
        return fail(ctx, NPBNN_E_STATE, "chain_run: a network on the weight-streamed path carries one weight set per pass (no group pass)");
    // (row shards: a gather between pass and step; weight-streamed path: the candidate image is patched between step and pass)
    int schedule = group_blocks > 0 ? NPBNN_SCHED_OVERLAP : (sharded || lp.wide) ? NPBNN_SCHED_SERIAL : cfg->schedule;
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
        // decides in a few microseconds: it leads where the predicted cost per decided pass says so (npbnn_ctx.hip.h, kSpecTurnExtraUs).
        // (its step workgroup must keep pace with the evaluation: with proposals much wider than a few hundred weights it does not)
        if (alone_on_device && seg_len == 0 && !ctx->sync_failed && ctx->persist_option && p_acc > 0.0 && group_blocks == 0 &&
            !cfg->slope_idx && M <= kPersistSerialMaxWidth && lp.fn_spec != nullptr &&
            (cfg->prior_kind == NPBNN_PRIOR_UNIFORM || (cfg->prior_kind == NPBNN_PRIOR_NORMAL && !cfg->prior_scale_w))) {
            // What decides is what a form was MEASURED to cost per iteration on this chain (npbnn_ctx.it_us: a batch's time over its
            // iterations); the model - a decided pass costs (1 + a) turns overlapped, a turn + `extra` with the decision between the
            // passes - only prices the form that has not run yet from the one that has.
            const double a = 1.0 - std::pow(1.0 - p_acc, D);                 // share of the passes that accept something
            const double ipp = a / p_acc;                                    // iterations a decided pass settles: 1 + (1-p) + ... + (1-p)^(D-1)
            const double extra = kSpecTurnExtraUs + kSpecTurnExtraUsPerWeight * M;
            const int kc = K < kShortBatch ? 0 : 1;                          // (a form's fixed cost per batch differs: short and long batches apart)
            double c_over = ctx->it_us[0][kc], c_ser = ctx->it_us[1][kc];    // us per iteration
            if (c_over <= 0.0 && c_ser <= 0.0) c_over = kTurnUsGuess * (1.0 + a) / ipp;
            if (c_over <= 0.0) {
                const double t_over = c_ser * ipp - extra > 5.0 ? c_ser * ipp - extra : 5.0;
                c_over = t_over * (1.0 + a) / ipp;
            }
            if (c_ser <= 0.0) c_ser = (c_over * ipp / (1.0 + a) + extra) / ipp;
            // (3 % in favour of the form the chain is on: no flipping on noise)
            const bool on_serial = ctx->last_schedule == NPBNN_SCHED_PERSIST_SERIAL;
            if (c_ser * (on_serial ? 0.97 : 1.03) < c_over) schedule = NPBNN_SCHED_PERSIST_SERIAL;
            // a measured cost goes stale while the other form runs (the chain's acceptance rate moves, the box's clocks do): after
            // kTurnReprobeBatches batches on one form, one batch on the other - if it is within reach or was never run
            // (a form's first batch in a size class pays for the switch - other builds of the kernel, cold tables: it runs a second one
            // before its figure is held against the other's)
            if (ctx->it_n[0][kc] == 1) schedule = NPBNN_SCHED_PERSIST;
            else if (ctx->it_n[1][kc] == 1) schedule = NPBNN_SCHED_PERSIST_SERIAL;
            const int other = schedule == NPBNN_SCHED_PERSIST_SERIAL ? 0 : 1;
            // (a form never measured in this size class gets its batch after a handful on the other: the model is a prior, not a verdict -
            // it prices every accept of the overlapped form at a whole pass in vain, and such a pass is cut short)
            const int since = ctx->turn_batches[other][kc];
            const bool never = ctx->it_us[other][kc] <= 0.0;
            if (since >= (never ? kTurnFirstProbeBatches : kTurnReprobeBatches)) {
                const double ratio = other == 1 ? c_ser / c_over : c_over / c_ser;
                // A form that has run: within a factor two, that is - its figure dates from when it last ran, the chain's acceptance rate has
                // moved since, and with it both forms' costs; one batch in kTurnReprobeBatches costs a per cent at worst.  A form never
                // run: when the model puts it within kFirstProbeWithin of the running one (at a few per cent acceptance the decision between
                // the passes is predicted a fifth to a third dearer, and is: two batches on it would be two batches lost - a chain's first
                // dispatches are where short runs are timed) - and whatever the model says after kTurnForcedProbeBatches, so that a wrong
                // model cannot park a chain for good.
                const bool probe = never ? (ratio < kFirstProbeWithin || since >= kTurnForcedProbeBatches) : ratio < 2.0;
                if (probe || !never) ctx->turn_batches[other][kc] = 0;
                if (probe) schedule = other == 1 ? NPBNN_SCHED_PERSIST_SERIAL : NPBNN_SCHED_PERSIST;
            }
        }
    }
    if ((schedule == NPBNN_SCHED_OVERLAP2 || schedule == NPBNN_SCHED_PERSIST) && ctx->sync_failed) schedule = NPBNN_SCHED_OVERLAP;
    // (asked for by name where it cannot run - no build with the speculative step for this launch, another prior, trainable slopes, a
    // shared GPU: the persistent overlapped form where that can, else kernel boundaries)
    // (the evaluating workgroups' tagged sums of that schedule sit in kSpecPartSlots slots per value, and the step reads exactly that
    // many: a part with more compute units than slots - none today: 256 of each on gfx950 - stays on the overlapped form)
    const int eval_wgs = lp.grid > ctx->n_cu - 1 ? ctx->n_cu - 1 : lp.grid;
    if (schedule == NPBNN_SCHED_PERSIST_SERIAL &&
        (eval_wgs > kSpecPartSlots || ctx->sync_failed || !alone_on_device || seg_len > 0 || group_blocks > 0 || lp.fn_spec == nullptr || cfg->slope_idx || getenv("NPBNN_NO_SPEC_STEP") ||
         !(cfg->prior_kind == NPBNN_PRIOR_UNIFORM || (cfg->prior_kind == NPBNN_PRIOR_NORMAL && !cfg->prior_scale_w))))
        schedule = (alone_on_device && seg_len == 0 && !ctx->sync_failed && group_blocks == 0) ? NPBNN_SCHED_PERSIST : NPBNN_SCHED_OVERLAP;
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
    const bool spec = pserial;             // (every condition was checked where the schedule was fixed)
    if (spec) lp.fn = lp.fn_spec;          // (the builds that carry spec_rounds)      // prepare the next pass ahead for every outcome
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
            ctx->d_spec_touch = nullptr; ctx->spec_touch_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_spec_touch, (size_t)kMaxCand * ctx->n_weights * 4 * sizeof(unsigned)));
            ctx->spec_touch_cap = (size_t)ctx->n_weights;
            ctx->spec_gen = 0xf0000000u;         // (forces the clearing below)
        }
        const size_t part_bytes = (size_t)2 * kMaxCand * kPartialStride * kSpecPartSlots * 2 * sizeof(unsigned long long);
        if (!ctx->d_spec_part) {
            HIP_TRY(ctx, hipMalloc(&ctx->d_spec_part, part_bytes));
            ctx->spec_gen = 0xf0000000u;         // (cleared below)
        }
        if (ctx->spec_gen + (unsigned)K + 8u >= 0xf0000000u) {   // the batch's pass tags (one per pass, at most K + 1 passes) could repeat
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_spec_touch, 0, (size_t)kMaxCand * ctx->spec_touch_cap * 4 * sizeof(unsigned), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_spec_part, 0, part_bytes, ctx->stream));
            ctx->spec_gen = 0;
        }
    }
    const size_t need = (size_t)K * M;
    if (need > ctx->draw_cap) {
        const size_t cap = need > (size_t)kChainMinCapacity * M ? need : (size_t)kChainMinCapacity * M;
        if (ctx->d_idx) (void)hipFree(ctx->d_idx);          // (d_delta lives behind it)
        if (ctx->d_pos) (void)hipFree(ctx->d_pos);
        if (ctx->d_pscale) (void)hipFree(ctx->d_pscale);
        ctx->d_idx = nullptr; ctx->d_delta = nullptr; ctx->d_pos = nullptr; ctx->d_pscale = nullptr; ctx->draw_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_idx, cap * (sizeof(int) + sizeof(double)) + 512));     // (indices, then the deviates: see below)
        HIP_TRY(ctx, hipMalloc(&ctx->d_pos, cap * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_pscale, cap * sizeof(float)));
        ctx->draw_cap = cap;
    }
    ctx->d_delta = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->d_idx) + ((size_t)K * M * sizeof(int) + 255) / 256 * 256);
    hipStream_t st = ctx->stream;
    double t_stage = wall_us();
    stage_mark(ctx, "(host set-up so far)", &t_stage);
    if (mask_packed) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mask, mask_packed, wb, hipMemcpyHostToDevice, st));
    // indices and deviates: ONE copy when the caller laid them out the way the device holds them (idx, then delta at the next
    // multiple of 256 bytes - npbnn_amd's pre-draw does), else one each
    const size_t idx_bytes = need * sizeof(int), delta_off = (idx_bytes + 255) / 256 * 256;
    if (reinterpret_cast<const char*>(delta) == reinterpret_cast<const char*>(idx) + delta_off) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_idx, idx, delta_off + need * sizeof(double), hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_idx, idx, idx_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_delta, delta, need * sizeof(double), hipMemcpyHostToDevice, st));
    }
    stage_mark(ctx, "copy of indices + deviates", &t_stage);
    const bool f16 = ctx->net.l0_f16 != 0;
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
    stage_mark(ctx, "copy of state + weights", &t_stage);
    // activation slopes of the batch: fixed ones (ActFun("genReLU", prm=...): cfg->cur_slopes with no slope draws) go into the network
    // description every candidate reads; trainable ones travel per candidate in the weight image (slope_idx below)
    for (int l = 0; l < kMaxLayers; ++l)
        ctx->net.act_prm[l] = (!cfg->slope_idx && l < cfg->n_slopes && l < NPBNN_MAX_LAYERS) ? (float)cfg->cur_slopes[l] : 0.f;
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
    c.partials = sharded ? ctx->d_shard_part : ctx->d_partials;
    c.image = ctx->d_image;
    // (weight-streamed path: the step keeps the candidate image itself while a proposal fits its staged entries; wider ones get two
    // launches over all compute units between step and pass - wide_cand_sync)
    c.cand_image = (lp.wide && M <= kWideStepPatchMax && D == 1) ? ctx->d_wide_cand : nullptr;
    // ... and where those launches run anyway they also MAKE the candidates (values, prior terms: wide_cand_prepare_kernel) - one
    // workgroup's walk over 3 x 2.6 k entries is 24 us of a 37-us step.  Plain batches without slopes of their own only (an exchange
    // run stops a chain at the proposal that leaves the fp16 range: the step must see that flag when it prepares).
    c.prep_terms = nullptr;
    B->make_cands = false;
    if (lp.wide && !c.cand_image && seg_len == 0 && !cfg->slope_idx && !getenv("NPBNN_WIDE_STEP_MAKES")) {
        const size_t need_terms = (size_t)kMaxCand * M;
        if (need_terms > ctx->prep_cap) {
            if (ctx->d_prep_terms) (void)hipFree(ctx->d_prep_terms);
            ctx->d_prep_terms = nullptr; ctx->prep_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_prep_terms, need_terms * sizeof(double)));
            ctx->prep_cap = need_terms;
        }
        c.prep_terms = ctx->d_prep_terms;
        B->make_cands = true;
    }
    c.pos = ctx->d_pos;
    c.pscale = f16 ? ctx->d_pscale : nullptr;
    c.pv = spec ? ctx->d_spec_pv : ctx->d_pv;      // (spec: the first step writes pass 0 into slot (parity 0, outcome 0))
    c.spec = nullptr;
    c.spec_pv = nullptr;
    c.spec_touch = nullptr;
    c.spec_part = nullptr;
    c.n_weights_spec = ctx->n_weights;
    c.spec_gen = 0;
    if (spec) {
        c.spec = ctx->d_spec;
        c.spec_pv = ctx->d_spec_pv;
        c.spec_touch = ctx->d_spec_touch;
        c.spec_part = ctx->d_spec_part;
        c.spec_gen = (int)ctx->spec_gen;
        ctx->spec_gen += (unsigned)K + 8u;             // (a pass decides at least one iteration)
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
    c.n_blocks = sharded ? ctx->shard_n : lp.n_waves;      // (row shards: one record per rank, gathered)
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
    c.n_rows = sharded ? ctx->shard_rows_total : d.n_rows;
    c.net = ctx->net;
    EvalParams p = make_params(ctx, d);
    p.partials = ctx->d_partials;
    p.inst_w = d.inst_w;
    p.use_classw = ctx->n_classw > 0 ? 1 : 0;
    p.has_pass = 1;
    if (getenv("NPBNN_EVAL_STAMPS") && group_blocks == 0) {
        HIP_TRY(ctx, hipMalloc(&B->d_estamps, (size_t)(lp.grid + 1) * 32 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemset(B->d_estamps, 0, (size_t)(lp.grid + 1) * 32 * sizeof(unsigned long long)));
        p.stamps = B->d_estamps;
    }
    p.pv = spec ? ctx->d_spec_pv : ctx->d_pv;
    p.pos = ctx->d_pos;
    p.pscale = f16 ? ctx->d_pscale : nullptr;
    p.M = M;
    p.chain = overlap ? ctx->d_cparams : nullptr;
    p.sync_mode = spec ? 3 : sync ? 1 : 0;
    // (the persistent overlapped launch: workgroups run up to a pass ahead of each other, so the heavier shares taking turns evens out)
    p.share_rot = (persist && !spec && group_blocks == 0 && lp.grid > 1 && !getenv("NPBNN_NO_SHARE_ROTATION")) ? d.n_tiles % lp.grid : 0;
    p.cand_slopes = c.slopes ? &ctx->d_slopes->cand[0][0][0] : nullptr;
    c.class_w = ctx->n_classw ? ctx->d_classw : nullptr;
    c.w_scale = f16 ? ctx->d_wscale : nullptr;
    {   // both parameter blocks in one copy (they sit in one device allocation, laid out like the staging area)
        memcpy(ctx->h_params, &p, sizeof(EvalParams));
        memcpy(ctx->h_params + sizeof(EvalParams) + sizeof(FinalizeParams), &c, sizeof(ChainParams));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eparams, ctx->h_params, sizeof(EvalParams) + sizeof(FinalizeParams) + sizeof(ChainParams),
                                    hipMemcpyHostToDevice, st));
    }
    // step (pack the weight image, prepare candidates) -> [eval -> step (decide + prepare)]* ; a pass consumes 1..D iterations, so
    // the number of passes is only known on the device
    stage_mark(ctx, "copy of parameter blocks", &t_stage);
    {   // one launch: image position (and fp16-split scale) of every drawn entry, so that no kernel needs a dependent lookup - indices
        // outside the network are neutralised there and reported when the batch comes back (kFlagBadIndex) - and, in the blocks
        // behind those, the weight image of the state the batch starts from (accepted candidates are committed to it entry by entry)
        const int gather_blocks = (int)((need + 255) / 256), pack_blocks = lp.wide ? 0 : (pack_item_count(ctx->net, true) + 255) / 256;
        PackJob pk;
        pk.w = lp.wide ? nullptr : ctx->d_wcur;      // (weight-streamed path: its own packing kernel, below)
        pk.class_w = ctx->n_classw ? ctx->d_classw : nullptr;
        pk.image = ctx->d_image;
        pk.net = reinterpret_cast<const NetMeta*>(reinterpret_cast<const char*>(ctx->d_cparams) + offsetof(ChainParams, net));
        pk.w_scale = f16 ? ctx->d_wscale : nullptr;
        hipLaunchKernelGGL(gather_pos_kernel, dim3((unsigned)(gather_blocks + pack_blocks)), dim3(256), 0, st, ctx->d_idx, (long long)need, ctx->n_weights,
                           (const int*)ctx->d_w2img, (const float*)(f16 ? ctx->d_w2scale : nullptr), ctx->d_pos,
                           f16 ? ctx->d_pscale : (float*)nullptr, ctx->d_chain_ovf, gather_blocks, pk);
    }
    if (lp.wide) {       // the committed image of the state the batch starts from, and the candidate image as a copy of it
        wide_pack(ctx, ctx->d_wcur, nullptr, ctx->d_image, ctx->d_chain_ovf);
        rc = wide_cand_begin(ctx);
        if (rc) return rc;
    }
    stage_mark(ctx, "gather + pack kernel", &t_stage);
    hipLaunchKernelGGL(chain_step_kernel, dim3(1), dim3(1024), 0, st, (const ChainParams*)ctx->d_cparams, 1);
    stage_mark(ctx, "first step kernel", &t_stage);
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
            if (lp.wide) {      // the layers' products and the likelihood of the candidate image (kept by the step, or by wide_cand_sync)
                if (B.M > kWideStepPatchMax || B.D > 1) wide_cand_sync(ctx, B.M, B.D, B.make_cands);
                const int rcw = wide_forward(ctx, 0, ctx->d_wide_cand, true, false, nullptr, B.D);
                if (rcw) return rcw;
            } else {
                hipLaunchKernelGGL(lp.fn, dim3(lp.grid), dim3(lp.wpb * 64), lp.lds, st, (const EvalParams*)ctx->d_eparams, 0, 1);
            }
            if (ctx->shard_n > 0) {
                const int rc = shard_exchange(ctx, lp, B.D);
                if (rc) return rc;
            }
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
    ctx->last_schedule = B.schedule;
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
            bool whole = true;          // (a step that skipped a phase - a void pass has nothing to reduce - left an older stamp there)
            for (int k = 1; k <= 6; ++k) whole = whole && q[k] >= q[k - 1];
            if (!whole) continue;
            for (int k = 1; k <= 6; ++k) acc[k] += (double)(q[k] - q[k - 1]) * 0.01;   // 100 MHz wall clock -> us
            ++n;
        }
        if (n) fprintf(stderr, "[npbnn step stamps] prefetch %.2f  reduce %.2f  decide %.2f  commit %.2f  prepare %.2f  finish %.2f us (mean of %d)\n",
                       acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, n);
        {   // start of a step -> start of the next one (the period of a pass where the step is what the passes wait for), and what of it the
            // step workgroup spent between two steps (waiting for the evaluating workgroups' sums, flag-ordered schedules).  Rows are
            // indexed by the first iteration of the decided pass: in time order
            std::vector<std::pair<unsigned long long, unsigned long long>> se;
            for (int r = 1; r < 1024; ++r) {
                const unsigned long long* q = &hs[(size_t)r * 8];
                if (q[0] && q[6] && q[6] > q[0]) se.push_back({q[0], q[6]});
            }
            std::sort(se.begin(), se.end());
            double period = 0, idle = 0;
            int m = 0;
            for (size_t i = 0; i + 1 < se.size(); ++i) {
                const double d = (double)(se[i + 1].first - se[i].first) * 0.01;
                if (se[i + 1].first < se[i].second || d > 200.0) continue;      // (a row overwritten by a later lap of the table; a batch boundary)
                period += d;
                idle += (double)(se[i + 1].first - se[i].second) * 0.01;
                ++m;
            }
            if (m) fprintf(stderr, "[npbnn step stamps] step start -> next step start %.2f us, of which between steps %.2f us (mean of %d)\n", period / m, idle / m, m);
        }
    }
    if (B.d_estamps) {
        report_eval_stamps(B.d_estamps, B.lp.grid + (B.overlap ? 1 : 0), B.lp.wpb, (B.sync || B.persist) ? 1 : 0);
        B.d_estamps = nullptr;
    }
    const int flags = *reinterpret_cast<const int*>(ctx->h_res + 448);
    if (flags & kFlagStructure) return fail(ctx, NPBNN_E_ARG, "chain_run: a layer-0 weight is not zero where the mask given to npbnn_set_layer_mask is");
    if (flags & kFlagBadIndex) return fail(ctx, NPBNN_E_ARG, "chain_run: a weight index of the batch is outside 0 .. %d (the entry was ignored)", ctx->n_weights - 1);
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
    while (t_done < K) {       // launch the least number of passes that can finish, look at the counter, repeat if short
        ++n_rounds;
        const double ta = timing ? wall_us() : 0.0;
        const int n_launch = passes_for(ctx, B, K - t_done, 1.0);
        double t_stage = wall_us();
        rc = chain_enqueue(ctx, B, n_launch);
        if (!rc) rc = chain_join(ctx, B);
        if (rc) return rc;
        HIP_TRY(ctx, hipGetLastError());
        stage_mark(ctx, "passes", &t_stage);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_res, ctx->d_res, B.RL.total, hipMemcpyDeviceToHost, st));   // state + results, one copy
        stage_mark(ctx, "copy of the results", &t_stage);
        const double tb = timing ? wall_us() : 0.0;
        if (ctx->shard_comm && ctx->shard_n > 1) {      // (collectives in the stream: a bounded wait - a peer that is gone must not hang this rank)
            if (npbnn_comm_wait_stream_(ctx->shard_comm, st, "the row-sharded chain batch") != 0)
                return fail(ctx, NPBNN_E_COMM, "chain_run: the row-sharded batch did not complete (a peer failed?)");
        } else {
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        if (timing) {
            const double tc = wall_us();
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
    if (B.persist && n_rounds == 1 && result->n_passes + result->n_void_passes >= 8) {      // what a turn of this form takes here (NPBNN_SCHED_AUTO)
        double& t = ctx->turn_us[B.schedule == NPBNN_SCHED_PERSIST_SERIAL ? 1 : 0];
        double now_us = (tw2 - B.tw1) / (result->n_passes + result->n_void_passes);
        // one stalled call (the host descheduled, a collector pause: 80 ms were seen) must not price a form out for good - the form
        // that lost is not run again, so nothing would ever correct its estimate: a sample counts for at most 1.25 x the estimate
        if (t > 0.0 && now_us > 1.25 * t) now_us = 1.25 * t;
        t = t > 0.0 ? 0.75 * t + 0.25 * now_us : now_us;
        const int form = B.schedule == NPBNN_SCHED_PERSIST_SERIAL ? 1 : 0, kc = K < kShortBatch ? 0 : 1;
        double& c = ctx->it_us[form][kc];                  // ... and what an iteration cost on it, in batches of this size class
        double it_now = (tw2 - B.tw1) / K;
        int& n_seen = ctx->it_n[form][kc];
        if (n_seen == 1) c = it_now < c ? it_now : c;      // (the better of a form's first two batches)
        else {
            if (c > 0.0 && it_now > 1.25 * c) it_now = 1.25 * c;
            c = c > 0.0 ? 0.75 * c + 0.25 * it_now : it_now;
        }
        if (n_seen < 1000) ++n_seen;
        ctx->turn_batches[form][kc] = 0;
        ctx->turn_batches[1 - form][kc] += 1;
    }
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
        L.rec = L.state + up256((size_t)n_seg * NPBNN_XSTATE_DOUBLES * sizeof(double));
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
        if (J.out_state) memcpy(J.out_state, ctx->h_xbuf + XL[q].state, (size_t)n_seg * NPBNN_XSTATE_DOUBLES * sizeof(double));
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

int npbnn_set_row_shard(npbnn_ctx* ctx, npbnn_comm* comm, npbnn_gather_fn gather, void* user, int32_t rank, int32_t n_ranks, int64_t n_rows_total) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (n_ranks == 0) {
        ctx->shard_n = 0;
        ctx->shard_comm = nullptr;
        ctx->shard_gather = nullptr;
        return NPBNN_OK;
    }
    if (n_ranks < 0 || n_ranks > 4096 || rank < 0 || rank >= n_ranks || n_rows_total < 1)
        return fail(ctx, NPBNN_E_ARG, "set_row_shard: rank %d of %d, %lld rows in all", rank, n_ranks, (long long)n_rows_total);
    if (n_ranks > 1 && !comm && !gather) return fail(ctx, NPBNN_E_ARG, "set_row_shard: %d ranks need a communicator or a gather callback", n_ranks);
    if (comm) {
        int device = -1, crank = -1, cworld = 0;
        if (npbnn_comm_info_(comm, &device, &crank, &cworld) != 0) return fail(ctx, NPBNN_E_COMM, "set_row_shard: the communicator is not usable");
        if (crank != rank || cworld != n_ranks || device != ctx->device)
            return fail(ctx, NPBNN_E_ARG, "set_row_shard: communicator is rank %d of %d on device %d, asked for rank %d of %d on device %d", crank, cworld,
                        device, rank, n_ranks, ctx->device);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t rec = (size_t)kMaxCand * kPartialStride;
    if (ctx->d_shard_recv) { (void)hipFree(ctx->d_shard_recv); ctx->d_shard_recv = nullptr; }
    if (ctx->d_shard_part) { (void)hipFree(ctx->d_shard_part); ctx->d_shard_part = nullptr; }
    if (ctx->h_shard) { (void)hipHostFree(ctx->h_shard); ctx->h_shard = nullptr; }
    HIP_TRY(ctx, hipMalloc(&ctx->d_shard_recv, rec * n_ranks * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_shard_part, rec * n_ranks * sizeof(double)));
    HIP_TRY(ctx, hipMemset(ctx->d_shard_recv, 0, rec * n_ranks * sizeof(double)));       // (values not in use stay zero)
    HIP_TRY(ctx, hipMemset(ctx->d_shard_part, 0, rec * n_ranks * sizeof(double)));
    HIP_TRY(ctx, hipHostMalloc(&ctx->h_shard, rec * (n_ranks + 1) * sizeof(double)));
    memset(ctx->h_shard, 0, rec * (n_ranks + 1) * sizeof(double));
    ctx->shard_n = n_ranks;
    ctx->shard_rank = rank;
    ctx->shard_rows_total = n_rows_total;
    ctx->shard_comm = comm;
    ctx->shard_gather = comm ? nullptr : gather;
    ctx->shard_user = user;
    ctx->its_per_pass = 0.0;          // (the ranks must agree on the number of launches: start from the same history)
    ctx->accept_rate = -1.0;
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

}  // extern "C"
