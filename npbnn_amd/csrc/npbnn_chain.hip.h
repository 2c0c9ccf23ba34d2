// Partial-sum reduction, log-likelihood from the sums, and the device-resident Metropolis-Hastings chain step
// (part of the device code of the npBNN hot path, see npbnn_kernels.hip.h)
#pragma once
#include "npbnn_common.hip.h"
#include "npbnn_pack.hip.h"

namespace npbnn {

// ------------------------------------------------------------------------------------------------
// finalize: reduce the per-wave partials in a fixed order and form the log-likelihood
// ------------------------------------------------------------------------------------------------
struct FinalizeParams {
    const double* partials;
    int n_waves;
    int lik_kind;
    int k_targets;
    long long n_rows;
    double lik_temp;
    int sigma_given;
    double sigma[NPBNN_MAX_TARGETS];
    npbnn_eval_out* out;   // device
};

// Sum value v of every wave's partial record: wave (threadIdx>>6) of the block takes values v = wave, wave+nw, ...;
// lane l adds records l, l+64, ... in order, then a fixed butterfly.  No float atomics -> deterministic.
// The values of a partial record that mean something: the log-likelihood sum (0), or - Gaussian likelihood - the residual moments
// of the k target columns (1 .. k and 1 + NPBNN_MAX_TARGETS .. + k); item i of the 1 + 2k (resp. 1) -> its index in the record
__device__ __forceinline__ int partial_value_count(int lik_kind, int k_targets) { return lik_kind == NPBNN_LIK_GAUSS ? 1 + 2 * k_targets : 1; }
__device__ __forceinline__ int partial_value_index(int item, int k_targets) {
    return item <= k_targets ? item : 1 + NPBNN_MAX_TARGETS + (item - k_targets - 1);
}

__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int n_blocks, int n_items, int k_targets, double* tot /*LDS*/) {
    // partials are laid out [value][workgroup]; wave w of this block sums the items w, w+nw, ... (the values in use): each lane adds
    // workgroups lane, lane+64, ... in order, then a fixed butterfly.  No float atomics -> deterministic.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int it = wave; it < n_items; it += nw) {
        const int v = partial_value_index(it, k_targets);
        double s = 0.0;
        for (int b = lane; b < n_blocks; b += 64) s += partials[(size_t)v * n_blocks + b];
        s = butterfly_sum_f64(s);
        if (lane == 0) tot[v] = s;
    }
    __syncthreads();
}

// log-likelihood (and sigma / residual moments) from the reduced totals; one thread.
__device__ __forceinline__ void loglik_from_totals(const double* tot, int lik_kind, int k_targets, long long n_rows, double lik_temp,
                                                   int sigma_given, const double* sigma_in, npbnn_eval_out* o) {
    o->n_rows = n_rows;
    for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) { o->sigma[j] = 0; o->sum_r[j] = 0; o->sum_r2[j] = 0; }
    if (lik_kind == NPBNN_LIK_GAUSS) {
        // sum_j [ -N (0.5 log 2pi + log s_j) - S2_j / (2 s_j^2) ];  empirical s_j = population std of the residuals
        // (np.std, BNN_env.py:475-476; scipy.stats.norm.logpdf, BNN_lib.py:131)
        const double N = (double)n_rows;
        double ll = 0.0;
        for (int j = 0; j < k_targets; ++j) {
            const double S1 = tot[1 + j], S2 = tot[1 + NPBNN_MAX_TARGETS + j];
            double sg;
            if (sigma_given) sg = sigma_in[j];
            else {
                const double mean = S1 / N;
                sg = sqrt(S2 / N - mean * mean);
            }
            o->sigma[j] = sg; o->sum_r[j] = S1; o->sum_r2[j] = S2;
            ll += -N * (0.9189385332046727418 + log(sg)) - S2 / (2.0 * sg * sg);
        }
        o->loglik = lik_temp * ll;
    } else {
        // the plug-in count likelihoods ignore lik_temp (BNN_lik.py:5-66)
        o->loglik = (lik_kind >= NPBNN_LIK_POISSON && lik_kind <= NPBNN_LIK_NEGBIN_BASE10 ? 1.0 : lik_temp) * tot[0];
    }
}

#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) finalize_kernel(const FinalizeParams* __restrict__ fp) {
    const FinalizeParams& f = *fp;
    __shared__ double tot[kPartialStride];
    reduce_partials(f.partials, f.n_waves, partial_value_count(f.lik_kind, f.k_targets), f.k_targets, tot);
    if (threadIdx.x == 0) loglik_from_totals(tot, f.lik_kind, f.k_targets, f.n_rows, f.lik_temp, f.sigma_given, f.sigma, f.out);
}
#endif  // NPBNN_KERNELS_MAIN

// ------------------------------------------------------------------------------------------------
// device-resident Metropolis-Hastings chain: state and parameter blocks (the step itself is chain_step, below)
// ------------------------------------------------------------------------------------------------
struct ChainDev {          // device-resident chain state
    double logLik, logPrior;
    double sigma[NPBNN_MAX_TARGETS];
    double cand_logPrior[2][kMaxCand];   // log priors of the candidates of the two passes in the pipeline (by pass parity)
    int t;                  // iterations decided so far in this batch
    int n_accepted;
    int n_passes;           // evaluation passes that decided at least one iteration
    int void_launch;        // overlapped mode: launch whose pass was evaluated from a state that an accept has since replaced
    int n_void;             // such passes in this batch
    int seg_end;            // iterations the chain may decide for now: K, or the end of the current segment of an exchange run
    double temperature;     // MCMC._temperature; changed on the device by the swaps of an exchange run
    double logPrior_rep;    // log prior as MCMC._logPrior holds it: updated on accept only (logPrior is re-summed at segment starts)
    int seg_idx;            // exchange run: segments exchanged so far
    int poisoned;           // exchange run: some chain had not reached the end of a segment when it was exchanged; nothing runs after
    // flag-ordered overlapped schedule (EvalParams.sync_mode): launches alternate between two streams and overlap; what a kernel
    // boundary used to guarantee is guaranteed by these, written with release / read with acquire at agent scope
    int prepared;           // passes 0 .. prepared are ready to be evaluated (descriptor, patch values, committed image); the step of
                            // launch L-1 sets it to L when it is done - the step of launch L waits for that too (steps run in order)
    int aborted;            // a wait timed out: every kernel of the batch leaves at once, the host retries on one stream
    int started;            // highest launch whose step workgroup has begun (see sync_step_leave)
    int exchanged;          // exchange run on two streams: exchanges applied so far, raised at the very end of exchange_apply_kernel
    int done[4];            // done[L & 3]: evaluating workgroups of the launches L, L-4, L-8 ... that have finished (cumulative)
};

// Trainable activation slopes of a chain (ActFun(trainable=True), np_bnn/BNN_env.py:416-421,502-503): the accepted slopes, the
// share of ChainDev.logPrior that is theirs (log(r) * -sum(slopes) * r, r = 10 - zero until the chain's first accept, because
// MCMC.__init__ computes its prior without it), and the slopes of the candidates of the two passes in the pipeline.
struct SlopeState {
    double cur[kMaxLayers];
    double term;
    double cand[2][kMaxCand][kMaxLayers];
};
__host__ __device__ inline double slope_prior_term(const double* s, int n) {
    double sum = 0.0;
    for (int l = 0; l < n; ++l) sum += s[l];
    return 2.302585092994046 * -sum * 10.0;       // np.log(r) * -np.sum(prm_tmp) * r, r = 10 (BNN_env.py:419-420)
}

struct SpecState;
struct ChainParams {
    ChainDev* st;
    PassDesc* pass;            // [2] candidates of the passes in the pipeline, by pass parity (read by the evaluation kernel)
    double* w_cur;             // float64 master copy of the current weights
    const double* mask;        // or nullptr
    const int* idx;            // [K][M] pre-drawn packed-weight indices (-1: superseded entry)
    const double* delta;       // [K][M]
    const int* cnt;            // [K]
    const double* log_u;       // [K]
    const double* hastings;    // [K] or nullptr
    const double* sigma_mult;  // [K][k_targets] or nullptr: Gaussian likelihood with an estimated error parameter - the proposal of
                               // iteration t is evaluated with sigma' = ChainDev.sigma * sigma_mult[t] (BNN_env.py:435-442)
    unsigned char* out_acc;    // [K]
    double* out_ll;            // [K] proposed logLik
    double* out_lp;            // [K] proposed logPrior
    const double* partials;    // [2][candidate][kPartialStride][n_blocks], by pass parity
    float* image;              // fragment image of the current weights, read by the evaluation kernel
    float* cand_image;         // weight-streamed path (npbnn_wide.hip.h; serial schedule, one candidate per pass): the image its kernels read -
                               // the committed image with the pending proposal's entries patched in by the step itself (prepare), and put
                               // back when the proposal is rejected; nullptr on the resident path, whose kernels patch their LDS copies
    double* prep_terms;        // weight-streamed path, candidates kept by launches over all compute units: [kMaxCand][M] - the step leaves the
                               // making of the candidates (values, prior terms, image entries) to wide_cand_prepare_kernel, which runs on the
                               // whole chip between the step and the pass, and adds up the prior terms of a pass - one per entry, in the order
                               // it would have added them itself - when it decides that pass; nullptr: the step makes them (every other case)
    const int* pos;            // [K][M] image position of every pre-drawn entry; bit 31 set: fp16-split layer-0 entry (the
                               // low bits are the half index of the high part, low part 512 halfs later); kSkipPos: none
    const float* pscale;       // [K][M] fp16-split column scale of every pre-drawn entry, or nullptr
    double* pv;                // [2][kMaxCand][M] proposed values of the candidates of the passes in the pipeline
    int* overflow;             // set when a scaled weight leaves the fp16 range
    unsigned long long* stamps; // diagnostics only (NPBNN_STEP_STAMPS=1), else nullptr
    SlopeState* slopes;        // trainable activation slopes, or nullptr
    const int* slope_idx;      // [K] the slope iteration t proposes to move ...
    const double* slope_delta; // [K] ... and by how much (reflected at 0 and 1)
    int n_slopes, slope_term_in;   // slope_term_in: the log prior handed in at the start of the batch holds the term of the accepted slopes
    SpecState* spec;           // NPBNN_SCHED_PERSIST_SERIAL: outcome-speculative preparation (spec_round below), else nullptr
    double* spec_pv;           // [3][kSpecOutcomes][kMaxCand][M] candidate patch values per pass (mod 3) and outcome
    unsigned* spec_touch;             // [kMaxCand][n_weights][4] pass tag of the last candidate j that touched the weight, -, and the value it gave it
    unsigned long long* spec_part;    // the evaluating workgroups' sums under that schedule: two tagged words per value (spec_part_*, below)
    int n_weights_spec, spec_gen;     // spec_gen: pass tags of this batch start above it (never reused: the host clears the tables first)
    const double* class_w;     // class weights for the weight image (pack_item), or nullptr
    const float* w_scale;      // fp16-split column scales of layer 0, or nullptr
    int K, M, D, n_blocks;
    int sync_test_skip;        // tests only: the step of this launch never reports back (-1: none) - exercises the time-out path of
                               // the two-stream schedule
    int stop_on_overflow;      // exchange run: a proposal outside the fp16 range stops the chain before it (else: flag only, the
                               // caller discards the batch)
    int prior_kind;
    double prior_scale[kMaxLayers];
    double half_inv_s2[kMaxLayers];   // 0.5 / scale^2 (normal prior)
    const double* prior_scale_w;      // one scale per packed weight (hyper-priors with a scale per input node or per weight,
                                      // npBNN.sample_prior_scale, BNN_env.py:196-221), or nullptr: one per layer (above)
    double w_bound;
    double lik_temp;           // (the temperature is chain state: ChainDev)
    int sigma_given;           // Gaussian: 1 = use sigma_fixed, 0 = empirical
    double sigma_fixed[NPBNN_MAX_TARGETS];
    long long n_rows;
    NetMeta net;
};

constexpr int kSkipPos = 0x7fffffff;

__device__ __forceinline__ double log_prior_density(int kind, double w, double scale) {
    if (kind == NPBNN_PRIOR_CAUCHY) return -log(3.14159265358979323846 * scale * (1.0 + (w / scale) * (w / scale)));
    if (kind == NPBNN_PRIOR_LAPLACE) return -log(2.0 * scale) - fabs(w) / scale;
    return -0.5 * (w / scale) * (w / scale) - log(scale) - 0.9189385332046727418;
}

// change of the log prior density when an entry moves from `b` to `v` (scale sc); the normal prior needs no
// transcendental: -(v^2 - b^2) / (2 sc^2)
__device__ __forceinline__ double prior_delta(int kind, double v, double b, double sc) {
    if (kind == NPBNN_PRIOR_NORMAL) return -0.5 * (v * v - b * b) / (sc * sc);
    if (kind == NPBNN_PRIOR_LAPLACE) return -(fabs(v) - fabs(b)) / sc;
    return log((sc * sc + b * b) / (sc * sc + v * v));                  // Cauchy
}

// image position / fp16-split scale of every pre-drawn entry, gathered once per batch
#ifdef NPBNN_KERNELS_CHAIN
struct PackJob {              // what the packing blocks of gather_pos_kernel need (w == nullptr: nothing to pack)
    const double* w;
    const double* class_w;
    float* image;
    const NetMeta* net;       // (device copy: ChainParams::net of the batch)
    const float* w_scale;
};
__global__ void __launch_bounds__(256) gather_pos_kernel(int* __restrict__ idx, long long n, int n_weights, const int* __restrict__ w2img,
                                                         const float* __restrict__ w2scale, int* __restrict__ pos, float* __restrict__ pscale,
                                                         int* __restrict__ flags, int gather_blocks, PackJob pk) {
    if ((int)blockIdx.x >= gather_blocks) {      // the blocks behind the gather pack the weight image of the state the batch starts from
        if (pk.w) pack_item(((int)blockIdx.x - gather_blocks) * 256 + (int)threadIdx.x, pk.w, nullptr, pk.class_w, pk.image, *pk.net, true, pk.w_scale, flags);
        return;
    }
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int w = idx[i];
    if (w >= n_weights) {          // an index outside the network: the entry is dropped here, the batch reports it (kFlagBadIndex)
        atomicOr(flags, kFlagBadIndex);
        idx[i] = -1;
        w = -1;
    }
    pos[i] = w >= 0 ? w2img[w] : kSkipPos;
    if (pscale) pscale[i] = (w >= 0 && w2scale) ? w2scale[w] : 1.0f;
}
#endif  // NPBNN_KERNELS_CHAIN

// l0_rows: NetMeta::l0_rows of the image (where the low part of a layer-0 entry marked kPosCompact sits)
__device__ __forceinline__ void patch_image(float* image, int pos, float scale, double v, int l0_rows) {
    if (pos == 0x7fffffff) return;                   // an entry the image does not hold (outside the layer-0 block structure: it is 0)
    if (pos < 0) {                                   // fp16-split entry (layer 0, or layer 1 of NetMeta::l1_f16)
        const float wv = (float)(v * (double)scale);
        _Float16 hi, lo;
        split_f16(wv, hi, lo);
        _Float16* img16 = reinterpret_cast<_Float16*>(image);
        const int h = pos & 0x3fffffff;
        img16[h] = hi;
        img16[h + ((pos & kPosCompact) ? 32 * l0_rows : 512)] = lo;
    } else {
        image[pos] = (float)v;
    }
}
__device__ __forceinline__ void patch_global_image(const ChainParams& c, int pos, float scale, double v) {
    patch_image(c.image, pos, scale, v, c.net.l0_rows);
}
// an entry of the candidate image back to what the committed image holds (weight-streamed path: images without compact rows)
__device__ __forceinline__ void restore_image_entry(float* cand, const float* image, int pos) {
    if (pos == 0x7fffffff) return;
    if (pos < 0) {
        const _Float16* src = reinterpret_cast<const _Float16*>(image);
        _Float16* dst = reinterpret_cast<_Float16*>(cand);
        const int h = pos & 0x3fffffff;
        dst[h] = src[h];
        dst[h + 512] = src[h + 512];
    } else {
        cand[pos] = image[pos];
    }
}

// block-wide sum of one double per thread, fixed order; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* red /*LDS, >= 16 doubles*/) {
    v = butterfly_sum_f64(v);
    __syncthreads();                                  // `red` may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// the proposal of one pre-drawn entry on base value `base` (UpdateNormal, BNN_mcmc.py:64-67; mask, BNN_env.py:461-462) and its
// change of the log prior - ONE definition for every schedule, contraction off, so that their chains agree to the bit.  The normal
// prior with one scale per layer (the default) is two multiplies and a subtraction; everything else (Cauchy, Laplace, a scale per
// weight) goes through a function of its own: its logarithms would otherwise be expanded at every call site.
// NORMAL_ONLY: the caller has made sure that the prior is uniform or normal with one scale per layer (spec_rounds: the other
// priors' logarithms, at its two dozen call sites, would not fit beside an evaluation kernel's hot loop - such chains run on the
// other schedules).  Same arithmetic either way.
template <bool NORMAL_ONLY>
__device__ __forceinline__ double spec_entry(double w_bound, int prior_kind, bool per_weight_scale, double base, double d, double m, double scale_w,
                                             double his, double lsc, double& dlp) {
#pragma clang fp contract(off)      // (no fused multiply-adds: the same bits wherever this is inlined)
    double v = base + d;
    if (v > w_bound) v = w_bound - (v - w_bound);
    if (v < -w_bound) v = -w_bound + (-w_bound - v);
    v *= m;
    if constexpr (NORMAL_ONLY) {
        if (prior_kind == NPBNN_PRIOR_NORMAL) dlp -= (v * v - base * base) * his;
    } else if (prior_kind != NPBNN_PRIOR_UNIFORM) {
        if (per_weight_scale) dlp += prior_delta(prior_kind, v, base, scale_w);
        else if (prior_kind == NPBNN_PRIOR_NORMAL) dlp -= (v * v - base * base) * his;
        else dlp += prior_delta(prior_kind, v, base, lsc);
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// device-resident Metropolis-Hastings chain, speculative over D candidates per pass
//
// K iterations of MCMC.mh_step (BNN_env.py:381-532, default path: UpdateNormal proposals, BNN_mcmc.py:57-69); the host
// pre-draws the random numbers of the K iterations (npbnn_host.c), so the proposals are the reference's.  chain_step (one
// workgroup):
//   1. decide the candidates of an evaluated pass, in iteration order (fixed-order reduction of the per-workgroup partials
//      -> logLik', accept test (logPost' - logPost) * temperature + hastings >= log u, BNN_env.py:493-494) and stop at the first
//      accepted one: it is committed to W_cur and to the weight image; later candidates of the pass were computed from a
//      state that no longer exists and are simply dropped (their iterations are evaluated again);
//   2. prepare the next candidates: for j < D, W_cur[idx] + delta of iteration t+j, reflected at +-bound and masked
//      (BNN_mcmc.py:64-67, BNN_env.py:461-462), stored as a patch list (the evaluation kernel applies it to its LDS image);
//      logPrior' = logPrior + sum of per-entry prior changes (npBNN.calc_prior, BNN_env.py:180-194; full sum at batch start).
//
// Two schedules (StepPlan):
//   serial      eval(L) -> step(L) -> eval(L+1) ...: the step decides the pass just evaluated and prepares the next one.
//   overlapped  one launch per pass; the last workgroup of launch L runs the step for pass L-1 WHILE the other workgroups
//               evaluate pass L, whose candidates were prepared one launch earlier on the assumption that pass L-1 rejects
//               everything (true for ~91 % of the passes at the 3 % acceptance rate of config 2).  The step prepares pass
//               L+1.  When pass L-1 does accept, pass L was evaluated from a state that no longer exists: it is marked void,
//               never decided, and pass L+1 restarts right after the accepted iteration.  Either way every decision is made
//               in iteration order on sums computed from the true current state: the chain is the sequential one.
// ------------------------------------------------------------------------------------------------
struct StepPlan {
    int first;     // first launch of a batch: iteration counter to 0, no pass to decide
    int resum;     // sum the prior of the current state in full (start of a batch, start of a segment of an exchange run)
    int dec;       // parity of the pass to decide, or -1
    int fly;       // parity of the pass being evaluated while this step runs (overlapped schedule), or -1
    int out;       // parity of the pass to prepare
    int launch;    // launch index within the batch (overlapped schedule)
};
__device__ __forceinline__ StepPlan overlapped_plan(int launch) {
    StepPlan pl;
    pl.first = 0;
    pl.resum = 0;
    pl.dec = launch >= 1 ? ((launch - 1) & 1) : -1;
    pl.fly = launch & 1;
    pl.out = (launch + 1) & 1;
    pl.launch = launch;
    return pl;
}

struct StepShared {            // LDS scratch of chain_step
    double tot[kMaxCand][kPartialStride];
    double red[16];
    double red3[kMaxCand][16];
    npbnn_eval_out o;
    double s_lp;               // log prior of the state the next candidates start from
    int s_accepted, s_t, s_start, s_lim;
};

// WIDE (the stand-alone step kernel only - the serial schedule, the weight-streamed path): proposals of tens of thousands of entries are
// walked four entries per thread at a time, and the step keeps the candidate image (ChainParams::cand_image).  The step INSIDE the
// evaluation kernels is instantiated without it: that code beside the tile loop costs the tightest builds registers in the loop
// (tests/test_build_hot_loops.py; the three-candidate pass kernel of config 2 went from 25.3 to 27.8 us with it).
template <bool WIDE = false>
__device__ __forceinline__ void chain_step(const ChainParams& c_generic, const StepPlan pl, StepShared& sh) {
    // The step's parameter block does not change while a launch runs (what its pointers point at does): read through the CONSTANT
    // address space its fields are scalar loads the compiler may keep - through the generic reference every field was fetched again
    // after every store (it cannot rule out that the store hit the block), a dependent round trip each time: a patch value stored,
    // the block's pointers re-read, waited for - six times over in the prepare phase (8 of its 11 us), and all through the decision.
    typedef const __attribute__((address_space(4))) ChainParams ConstChainParams;
    ConstChainParams& c = *(ConstChainParams*)&c_generic;
    const int tid = threadIdx.x;
    ChainDev* st = c.st;
    const int lik_kind = c.net.lik_kind;
    const size_t pv_stride = (size_t)kMaxCand * c.M;
    const size_t part_stride = (size_t)kMaxCand * kPartialStride * c.n_blocks;
    const int stamp_row = pl.first ? 0 : (c.pass[pl.dec >= 0 ? pl.dec : 0].t0 & 1023);
#define NPBNN_STAMP(k) do { if (c.stamps && threadIdx.x == 0) c.stamps[(size_t)stamp_row * 8 + (k)] = wall_clock64(); } while (0)
    NPBNN_STAMP(0);

    // at the start of a batch the prior of the current state is summed in full (proposals then update it
    // incrementally from the touched entries, so rounding drift cannot accumulate across batches)
    if (pl.resum && c.prior_kind != NPBNN_PRIOR_UNIFORM) {
        double lp = 0.0;
        for (int l = 0; l < c.net.n_layers; ++l) {
            const auto& L = c.net.L[l];
            const int n = L.out_dim * (L.in_dim + L.has_bias);
            const double sc = c.prior_scale[l];
            if (c.prior_scale_w) {            // a scale per weight: the density entry by entry
                for (int i = tid; i < n; i += blockDim.x) lp += log_prior_density(c.prior_kind, c.w_cur[L.w_off + i], c.prior_scale_w[L.w_off + i]);
            } else if (c.prior_kind == NPBNN_PRIOR_NORMAL) {
                double q = 0.0;
                for (int i = tid; i < n; i += blockDim.x) { const double w = c.w_cur[L.w_off + i]; q += w * w; }
                lp += -0.5 * q / (sc * sc);
                if (tid == 0) lp -= (double)n * (log(sc) + 0.9189385332046727418);
            } else {
                for (int i = tid; i < n; i += blockDim.x) lp += log_prior_density(c.prior_kind, c.w_cur[L.w_off + i], sc);
            }
        }
        const double s = block_sum(lp, sh.red);
        if (tid == 0) st->logPrior = s;
    }
    if (pl.resum && c.slopes && tid == 0) {       // (after the re-sum; the uniform prior keeps the value handed in, which holds the term)
        const double term = c.slope_term_in ? slope_prior_term(c.slopes->cur, c.n_slopes) : 0.0;
        c.slopes->term = term;
        if (c.prior_kind != NPBNN_PRIOR_UNIFORM) st->logPrior += term;
    }

    // ---- 1. decide the pending candidates ----
    int t0 = 0, n_pend = 0;
    if (pl.dec >= 0) {
        t0 = c.pass[pl.dec].t0;
        n_pend = c.pass[pl.dec].n_cand;
        if (pl.fly >= 0 && st->void_launch == pl.launch - 1) n_pend = 0;     // that pass saw a state that an accept replaced
    }
    double prefetch_sink = 0.0;
    bool deferred = false;
    if constexpr (WIDE) deferred = c.prep_terms != nullptr;
    if (pl.fly < 0 && !deferred) {   // serial schedule: whichever candidate wins, the next pass starts at t0+1 .. t0+n_pend: pull those rows of
        // the pre-drawn arrays towards the L2 now, while the partial sums are being reduced (the values are not used here)
        const int r_lo = t0 + (pl.first ? 0 : 1), r_hi = min(c.K, t0 + n_pend + c.D);
        double sink = 0.0;
        for (int r = r_lo; r < r_hi; ++r)
            if (tid < c.M) sink += (double)c.idx[(size_t)r * c.M + tid] + c.delta[(size_t)r * c.M + tid] + (double)c.pos[(size_t)r * c.M + tid];
        prefetch_sink = sink;
    }
    // decision operands, fetched now by the deciding thread so that they are in registers when the sums arrive
    double d_cand[kMaxCand], d_logu[kMaxCand], d_h[kMaxCand], d_ll = 0.0, d_lp = 0.0, d_temp = 1.0;
#pragma unroll
    for (int j = 0; j < kMaxCand; ++j) { d_cand[j] = 0.0; d_logu[j] = 0.0; d_h[j] = 0.0; }
    if constexpr (WIDE) {
        if (deferred && n_pend > 0) {      // the prior terms wide_cand_prepare_kernel left, one per entry: thread, wave and workgroup add them in
                                           // the order the step's own preparation does (entry tid, tid + 1024, ...: the same bits)
            constexpr int TS = 3;                  // (a thread's first three terms of every candidate are requested together)
            int n_terms[kMaxCand];
            double x[kMaxCand][TS];
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j) {
                n_terms[j] = j < n_pend ? c.pass[pl.dec].cnt[j] : 0;
#pragma unroll
                for (int u = 0; u < TS; ++u) {
                    const int e = tid + u * (int)blockDim.x;
                    x[j][u] = e < n_terms[j] ? c.prep_terms[(size_t)j * c.M + e] : 0.0;
                }
            }
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j) {
                const double* tj = c.prep_terms + (size_t)j * c.M;
                double sum = 0.0;
#pragma unroll
                for (int u = 0; u < TS; ++u)
                    if (tid + u * (int)blockDim.x < n_terms[j]) sum += x[j][u];
                for (int e = tid + TS * (int)blockDim.x; e < n_terms[j]; e += (int)blockDim.x) sum += tj[e];
                sum = butterfly_sum_f64(sum);
                if ((tid & 63) == 0) sh.red3[j][tid >> 6] = sum;
            }
            __syncthreads();
        }
    }
    if (tid == 0 && n_pend > 0) {
        d_ll = st->logLik;
        d_lp = st->logPrior;
        d_temp = st->temperature;
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
            if (j < n_pend) {
                d_cand[j] = st->cand_logPrior[pl.dec][j];
                if constexpr (WIDE) {
                    if (deferred) {         // (held the log prior of the state the candidates start from: base + sum, as the step forms it)
                        double sj = 0.0;
                        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sj += sh.red3[j][w];
                        d_cand[j] = d_cand[j] + sj;
                    }
                }
                d_logu[j] = c.log_u[t0 + j];
                d_h[j] = c.hastings ? c.hastings[t0 + j] : 0.0;
            }
    }
    NPBNN_STAMP(1);
    if (n_pend > 0) {
        const int nvals = partial_value_count(lik_kind, c.net.k_targets);      // (only the values in use: 1, or 1 + 2k of the 33)
        {   // wave w sums items w, w+nw, ... (item = candidate * nvals + value): lanes add workgroups lane, lane+64, ... in order
            const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
            const double* part = c.partials + (size_t)pl.dec * part_stride;
            for (int item = wave; item < n_pend * nvals; item += nw) {
                const int j = item / nvals, v = partial_value_index(item % nvals, c.net.k_targets);
                const double* src = part + ((size_t)j * kPartialStride + v) * c.n_blocks;
                double x[4];           // (requested together, added in order: see spec_rounds)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b = lane + 64 * u;
                    x[u] = b < c.n_blocks ? src[b] : 0.0;
                }
                double s = 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (lane + 64 * u < c.n_blocks) s += x[u];
                for (int b = lane + 256; b < c.n_blocks; b += 64) s += src[b];
                s = butterfly_sum_f64(s);
                if (lane == 0) sh.tot[j][v] = s;
            }
        }
        __syncthreads();
        NPBNN_STAMP(2);
        if (tid == 0) {
            int accepted = -1, n_done = n_pend;
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j) {
                if (j < n_pend && accepted < 0) {
                    const int t = t0 + j;
                    if (c.sigma_mult) {
                        double sg[NPBNN_MAX_TARGETS];
                        for (int q = 0; q < c.net.k_targets; ++q) sg[q] = st->sigma[q] * c.sigma_mult[(size_t)t * c.net.k_targets + q];
                        loglik_from_totals(sh.tot[j], lik_kind, c.net.k_targets, c.n_rows, c.lik_temp, 1, sg, &sh.o);
                    } else {
                        loglik_from_totals(sh.tot[j], lik_kind, c.net.k_targets, c.n_rows, c.lik_temp, c.sigma_given, c_generic.sigma_fixed, &sh.o);
                    }
                    const double lp = d_cand[j];
                    const double post_new = sh.o.loglik + lp, post_old = d_ll + d_lp;
                    const int a = ((post_new - post_old) * d_temp + d_h[j] >= d_logu[j]) ? 1 : 0;
                    c.out_acc[t] = (unsigned char)a;
                    c.out_ll[t] = sh.o.loglik;
                    c.out_lp[t] = lp;
                    if (a) {
                        st->logLik = sh.o.loglik;
                        st->logPrior = lp;
                        st->logPrior_rep = lp;
                        st->n_accepted += 1;
                        if (lik_kind == NPBNN_LIK_GAUSS)
                            for (int q = 0; q < c.net.k_targets; ++q) st->sigma[q] = sh.o.sigma[q];
                        if (c.slopes) {
                            for (int l = 0; l < c.n_slopes; ++l) c.slopes->cur[l] = c.slopes->cand[pl.dec][j][l];
                            c.slopes->term = slope_prior_term(c.slopes->cur, c.n_slopes);
                        }
                        accepted = j;
                        n_done = j + 1;
                        sh.s_lp = lp;
                    }
                }
            }
            if (accepted < 0) sh.s_lp = d_lp;
            st->t = t0 + n_done;
            st->n_passes += 1;
            sh.s_accepted = accepted;
            sh.s_t = t0 + n_done;
            // where the next candidates start: right after the decided iterations - unless a pass is being evaluated right now
            // and is still good (nothing accepted): it covers the iterations after these, the new candidates follow it
            int start = t0 + n_done;
            if (pl.fly >= 0) {
                if (accepted >= 0) {
                    __hip_atomic_store(&st->void_launch, pl.launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (polled by the evaluating workgroups)
                    if (c.pass[pl.fly].n_cand > 0) st->n_void += 1;
                } else {
                    start = c.pass[pl.fly].t0 + c.pass[pl.fly].n_cand;
                }
            }
            sh.s_start = start;
            sh.s_lim = st->seg_end;
        }
        __syncthreads();
        NPBNN_STAMP(3);
        const int a = sh.s_accepted;
        if constexpr (WIDE) {
            constexpr int CU = 4;   // entries per thread requested together (wide proposals: each level of dependent loads is a round trip)
            if (a >= 0) {           // commit the accepted candidate: master weights and the global weight image
                const size_t row = (size_t)(t0 + a) * c.M;
                const int n = c.cnt[t0 + a];
                const double* pv = c.pv + (size_t)pl.dec * pv_stride;
                for (int e0 = tid; e0 < n; e0 += CU * (int)blockDim.x) {
                    int ci[CU], cp[CU];
                    double cv[CU];
                    float cs[CU];
    #pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        const int e = e0 + u * (int)blockDim.x;
                        ci[u] = -1; cp[u] = 0; cv[u] = 0.0; cs[u] = 1.0f;
                        if (e < n) {
                            ci[u] = c.idx[row + e];
                            cv[u] = pv[(size_t)a * c.M + e];
                            cp[u] = c.pos[row + e];
                            if (c.pscale) cs[u] = c.pscale[row + e];
                        }
                    }
    #pragma unroll
                    for (int u = 0; u < CU; ++u)
                        if (ci[u] >= 0) {
                            c.w_cur[ci[u]] = cv[u];
                            patch_global_image(c_generic, cp[u], cs[u], cv[u]);
                        }
                }
            } else if (c.cand_image) {      // rejected (weight-streamed path): the candidate image's patched entries back to the committed values
                const size_t row = (size_t)t0 * c.M;
                const int n = c.cnt[t0];
                for (int e0 = tid; e0 < n; e0 += CU * (int)blockDim.x) {
                    int cp[CU];
    #pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        const int e = e0 + u * (int)blockDim.x;
                        cp[u] = e < n ? c.pos[row + e] : 0x7fffffff;
                    }
    #pragma unroll
                    for (int u = 0; u < CU; ++u) restore_image_entry(c.cand_image, c.image, cp[u]);
                }
            }
        } else {
            if (a >= 0) {           // commit the accepted candidate: master weights and the global weight image
                const size_t row = (size_t)(t0 + a) * c.M;
                const int n = c.cnt[t0 + a];
                const double* pv = c.pv + (size_t)pl.dec * pv_stride;
                for (int e = tid; e < n; e += blockDim.x) {
                    const int i = c.idx[row + e];
                    if (i >= 0) {
                        const double v = pv[(size_t)a * c.M + e];
                        c.w_cur[i] = v;
                        patch_global_image(c_generic, c.pos[row + e], c.pscale ? c.pscale[row + e] : 1.0f, v);
                    }
                }
            }
        }
        __syncthreads();
    } else if (tid == 0) {
        const int t_now = pl.first ? 0 : st->t;
        if (pl.first) st->t = 0;
        sh.s_t = t_now;
        sh.s_lp = st->logPrior;
        // nothing decided (start of a batch, or the pending pass was void): the pass in flight, if any, is good
        sh.s_start = pl.fly >= 0 ? c.pass[pl.fly].t0 + c.pass[pl.fly].n_cand : t_now;
        sh.s_lim = st->seg_end;
    }
    __syncthreads();

    NPBNN_STAMP(4);
    // ---- 2. prepare the next candidates: each is the current state plus its own iteration's perturbation.  Work items
    //      are (candidate, entry) pairs spread over the whole workgroup; the three prior sums share one reduction. ----
    const int t_new = sh.s_start;
    int n_new = sh.s_lim - t_new;            // (seg_end <= K)
    if (n_new > c.D) n_new = c.D;
    if (n_new < 0) n_new = 0;
    if (n_new == 0) {        // nothing left to prepare (end of the batch or of a swap interval; launches enqueued past it): an
        if (tid == 0) {      // empty descriptor and out - these launches are pure overhead, keep them short
            PassDesc d;
            d.t0 = t_new;
            d.n_cand = 0;
            for (int j = 0; j < kMaxCand; ++j) d.cnt[j] = 0;
            // pad[0] = 1: terminal - every iteration the chain may decide for now has been decided, so nothing will be prepared
            // after this (an empty pass that is NOT terminal: the pass in flight reaches the limit but is still to be decided - if
            // it accepts, the iterations after the accepted one are proposed again).  The persistent launch ends on it.
            d.pad[0] = sh.s_t >= sh.s_lim ? 1 : 0;
            d.pad[1] = d.pad[2] = 0;
            c.pass[pl.out] = d;
        }
        return;
    }
    double dlp[kMaxCand];
    if constexpr (WIDE) {
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) dlp[j] = 0.0;
    }
    if (!deferred) {
        // staged so that the loads of all candidates are in flight together: (1) the pre-drawn entry, (2) the weight it
        // touches, (3) arithmetic and stores.  One entry per thread and candidate; wider proposals loop.
        const double* __restrict__ wcur = c.w_cur;
        const double* __restrict__ mask = c.mask;
        double* __restrict__ pv_out = c.pv + (size_t)pl.out * pv_stride;
        int woff[kMaxLayers];
        double half_inv_s2[kMaxLayers], layer_scale[kMaxLayers];
#pragma unroll
        for (int q = 0; q < kMaxLayers; ++q) {
            woff[q] = q < c.net.n_layers ? c.net.L[q].w_off : 0x7fffffff;
            half_inv_s2[q] = c.half_inv_s2[q];
            layer_scale[q] = c.prior_scale[q];
        }
        // ES entries per thread and candidate are staged (wider proposals than ES x the workgroup loop behind them)
        constexpr int ES = 2;
        int ii[kMaxCand][ES], pp[kMaxCand][ES], cn[kMaxCand];
        double dd[kMaxCand][ES], bb[kMaxCand][ES], mm[kMaxCand][ES], sw[kMaxCand][ES];
        float ss[kMaxCand][ES];
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
            dlp[j] = 0.0;
            cn[j] = j < n_new ? c.cnt[t_new + j] : 0;
#pragma unroll
            for (int u = 0; u < ES; ++u) {
                const int e = tid + u * (int)blockDim.x;
                ii[j][u] = -1; pp[j][u] = 0; dd[j][u] = 0.0; ss[j][u] = 1.0f;
                // (asked for with the row's count, not behind it: the rows are touched once per batch - every level of dependent loads here
                // is a cold miss of 2-3 us on the step's critical path; slots past the count are inside the [K][M] arrays and are dropped below)
                if (j < n_new && e < c.M) {
                    const size_t k = (size_t)(t_new + j) * c.M + e;
                    ii[j][u] = c.idx[k];
                    dd[j][u] = c.delta[k];
                    pp[j][u] = c.pos[k];
                    if (c.pscale) ss[j][u] = c.pscale[k];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
#pragma unroll
            for (int u = 0; u < ES; ++u)
                if (!(tid + u * (int)blockDim.x < cn[j])) { ii[j][u] = -1; pp[j][u] = 0; dd[j][u] = 0.0; ss[j][u] = 1.0f; }
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
#pragma unroll
            for (int u = 0; u < ES; ++u) {
                const int i = ii[j][u];
                bb[j][u] = i >= 0 ? wcur[i] : 0.0;
                mm[j][u] = (i >= 0 && mask) ? mask[i] : 1.0;
                sw[j][u] = (i >= 0 && c.prior_scale_w) ? c.prior_scale_w[i] : 1.0;
            }
        auto make = [&](int j, int e, int i, double base, double d, double m, int pos, float sc, double scale_w) {
            double his = half_inv_s2[0], lsc = layer_scale[0];     // (selected, not indexed: a register array indexed at run time
#pragma unroll                                                     //  goes to scratch)
            for (int q = 1; q < kMaxLayers; ++q) {
                const bool past = i >= woff[q];
                his = past ? half_inv_s2[q] : his;
                lsc = past ? layer_scale[q] : lsc;
            }
            const double v = spec_entry<false>(c.w_bound, c.prior_kind, c.prior_scale_w != nullptr, base, d, m, scale_w, his, lsc, dlp[j]);
            pv_out[(size_t)j * c.M + e] = v;
            if (pos < 0 && !(fabs(v * (double)sc) <= (double)kF16Safe)) atomicOr(c.overflow, kFlagF16Range);
            if constexpr (WIDE) { if (c.cand_image) patch_image(c.cand_image, pos, sc, v, 16); }      // (weight-streamed path: the image its kernels read)
        };
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
#pragma unroll
            for (int u = 0; u < ES; ++u)
                if (ii[j][u] >= 0) make(j, tid + u * (int)blockDim.x, ii[j][u], bb[j][u], dd[j][u], mm[j][u], pp[j][u], ss[j][u], sw[j][u]);
            if (cn[j] > ES * (int)blockDim.x) {
                if constexpr (WIDE) {
                    // (proposals wider than the staged entries: four entries per thread requested together, made in entry order - the
                    // order a thread adds its prior terms in is part of the sum's bits)
                    constexpr int PU = 4;
                    const size_t row = (size_t)(t_new + j) * c.M;
                    for (int e0 = tid + ES * (int)blockDim.x; e0 < cn[j]; e0 += PU * (int)blockDim.x) {
                        int qi[PU], qp[PU];
                        double qd[PU], qb[PU], qm[PU], qw[PU];
                        float qs[PU];
    #pragma unroll
                        for (int u = 0; u < PU; ++u) {
                            const int e = e0 + u * (int)blockDim.x;
                            qi[u] = -1; qp[u] = 0; qd[u] = 0.0; qs[u] = 1.0f;
                            if (e < cn[j]) {
                                qi[u] = c.idx[row + e];
                                qd[u] = c.delta[row + e];
                                qp[u] = c.pos[row + e];
                                if (c.pscale) qs[u] = c.pscale[row + e];
                            }
                        }
    #pragma unroll
                        for (int u = 0; u < PU; ++u) {
                            const int i = qi[u];
                            qb[u] = i >= 0 ? wcur[i] : 0.0;
                            qm[u] = (i >= 0 && mask) ? mask[i] : 1.0;
                            qw[u] = (i >= 0 && c.prior_scale_w) ? c.prior_scale_w[i] : 1.0;
                        }
    #pragma unroll
                        for (int u = 0; u < PU; ++u)
                            if (qi[u] >= 0) make(j, e0 + u * (int)blockDim.x, qi[u], qb[u], qd[u], qm[u], qp[u], qs[u], qw[u]);
                    }
                } else {
                    const size_t row = (size_t)(t_new + j) * c.M;
                    for (int e = tid + ES * blockDim.x; e < cn[j]; e += blockDim.x) {
                        const int i = c.idx[row + e];
                        if (i >= 0) make(j, e, i, wcur[i], c.delta[row + e], mask ? mask[i] : 1.0, c.pos[row + e], c.pscale ? c.pscale[row + e] : 1.0f,
                                         c.prior_scale_w ? c.prior_scale_w[i] : 1.0);
                    }
                }
            }
        }
    }
    NPBNN_STAMP(5);
#pragma unroll
    for (int j = 0; j < kMaxCand; ++j) {
        dlp[j] = butterfly_sum_f64(dlp[j]);
        if ((tid & 63) == 0) sh.red3[j][tid >> 6] = dlp[j];
    }
    __syncthreads();
    if (tid == 0) {
        const double base_lp = sh.s_lp;
        if (c.stop_on_overflow && n_new > 0 && (atomicAdd(c.overflow, 0) & kFlagF16Range) != 0) {     // (the flag was raised before the barrier above)
            n_new = 0;
            st->poisoned = 1;
        }
        PassDesc d;
        d.t0 = t_new;
        d.n_cand = n_new;
        for (int j = 0; j < kMaxCand; ++j) {
            double sj = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sj += sh.red3[j][w];
            if (c.slopes && j < n_new) {              // UpdateNormal1D(acc_prm, d, n = 1, Mb = 1, mb = 0) with the pre-drawn entry and step
                const int k = c.slope_idx[t_new + j];
                const double step = c.slope_delta[t_new + j];
                double sum = 0.0;
                for (int l = 0; l < kMaxLayers; ++l) {       // (no local array: a run-time index would put it in scratch)
                    double v = l < c.n_slopes ? c.slopes->cur[l] : 0.0;
                    if (l == k) {
                        v += step;
                        if (v > 1.0) v = 1.0 - (v - 1.0);
                        if (v < 0.0) v = 0.0 + (0.0 - v);
                    }
                    c.slopes->cand[pl.out][j][l] = v;
                    if (l < c.n_slopes) sum += v;
                }
                sj += 2.302585092994046 * -sum * 10.0 - c.slopes->term;      // (slope_prior_term of the candidate)
            }
            if (j < n_new) st->cand_logPrior[pl.out][j] = base_lp + sj;
            d.cnt[j] = j < n_new ? c.cnt[t_new + j] : 0;
        }
        d.pad[0] = d.pad[1] = d.pad[2] = 0;
        c.pass[pl.out] = d;
    }
    if (prefetch_sink == 1.2345e300) c.out_lp[0] = prefetch_sink;      // keeps the prefetch loads alive; never true
    NPBNN_STAMP(6);
#undef NPBNN_STAMP
}

// ---- flag-ordered overlapped schedule: waits and signals (all at agent scope; tools/microbench_handoff.hip measures the
// hand-over: a flag is seen 0.6 us after it was raised, data written before a release store is fresh after an acquire) ----
// 250 ms of the 100 MHz wall clock.  A legitimate wait is < 0.1 ms, but the GPU can stand still for longer than that for reasons that
// have nothing to do with the protocol (at 10 ms the million-iteration test saw one time-out in roughly every seventh run of the whole
// suite and none in any run on its own); a wait that really cannot end - a workgroup that is not resident - is still found, a quarter
// of a second later, and the batch repeated on kernel boundaries.
constexpr unsigned long long kSyncTimeoutTicks = 25000000ull;

__device__ __forceinline__ bool sync_wait_ge(ChainDev* st, const int* word, int target, unsigned long long timeout_ticks = kSyncTimeoutTicks) {       // one thread
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__hip_atomic_load(&st->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(&st->aborted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return !__hip_atomic_load(&st->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// evaluating workgroup of launch L: its pass must have been prepared (by the step of launch L-1)
__device__ __forceinline__ bool sync_eval_enter(ChainDev* st, int launch, int* lds_flag, int early_prepared) {
    // No acquire fence: at agent scope it drops this XCD's whole L2, and workgroups of overlapping launches arrive at scattered
    // times - every arrival would throw out lines its 31 neighbours are about to use (measured: passes of 45-55 us).  What the
    // step hands over is READ coherently instead: descriptor and patch values with agent-scope loads, the weight image with
    // agent-scope LDS-DMA (dma16_coherent) - the lines come from the memory side whatever the caches hold.
    // `early_prepared`: the flag as thread 0 read it at the very top of the kernel (the round trip hides under the parameter
    // loads); nearly always it already says "ready".
    if (threadIdx.x == 0) {
        int ok = 1;
        // (ready already: no look at ChainDev.aborted - a round trip per pass - here: a pass that is prepared may be evaluated whatever
        // happened elsewhere, and a batch that was aborted never prepares the next one: the wait for that one sees the flag)
        if (early_prepared < launch || early_prepared == 0x7fffffff) ok = sync_wait_ge(st, &st->prepared, launch) ? 1 : 0;
        *lds_flag = ok;
    }
    __syncthreads();
    const int ok = *lds_flag;
    __syncthreads();            // the word sits where the weight image is about to land: nobody copies before everybody has read it
    return ok != 0;
}
__device__ __forceinline__ void sync_eval_leave(ChainDev* st, int launch) {      // one thread, after a barrier behind the workgroup's last
    __builtin_amdgcn_s_waitcnt(0);                                               // (agent-scope, write-through) store of its sums
    __hip_atomic_fetch_add(&st->done[launch & 3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// step of launch L: the step of launch L-1 must be through, and every workgroup that evaluated pass L-1 must have left
__device__ __forceinline__ bool sync_step_enter(const ChainParams& c, int launch, int n_eval_wgs, int* lds_flag) {
    ChainDev* st = c.st;
    if (c.sync_test_skip == launch) return false;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&st->started, launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = sync_wait_ge(st, &st->prepared, launch);
        if (ok && launch >= 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int slot = (launch - 1) & 3;
            ok = sync_wait_ge(st, &st->done[slot], ((launch - 1) / 4 + 1) * n_eval_wgs);
        }
        *lds_flag = ok ? 1 : 0;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return *lds_flag != 0;
}
// `wait_for_next`: launch next_launch has been enqueued and this step belongs to launch next_launch - 1.  The step then stays
// until the step workgroup of launch next_launch has begun.  Why: launch L+1 starts when launch L-1 (same stream) is complete,
// its evaluating workgroups wait for step L - so step L must hold a compute unit before they can take them all.  Launch L
// normally starts first (it follows launch L-2), but short launches (void, empty) can finish out of order; with this wait
// launch L-1 cannot complete, hence launch L+1 cannot start, before step L is resident.
__device__ __forceinline__ void sync_step_leave(ChainDev* st, int next_launch, bool wait_for_next = false) {     // whole workgroup
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&st->prepared, next_launch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (wait_for_next) (void)sync_wait_ge(st, &st->started, next_launch);
    }
}

// ------------------------------------------------------------------------------------------------
// NPBNN_SCHED_PERSIST_SERIAL: evaluate pass P, decide it, evaluate pass P + 1 from the state it left (the order of the reference's
// loop, BNN_env.py:449-494) inside ONE persistent launch - with everything the decision does NOT need done beforehand.
// While the evaluating workgroups are busy with pass P (candidates of iterations t0 .. t0+n-1 on the current state S), the step
// workgroup prepares the next pass for EVERY outcome pass P can have: nothing accepted (next candidates t0+n.. on S) or candidate
// j accepted (next candidates t0+j+1.. on S + its perturbation) - n + 1 outcomes, up to D candidates each: patch lists and log
// priors.  The accepted states themselves are never materialised ahead: a candidate entry of an accepting outcome takes its base
// value from candidate j's patch list where j touched that weight (`touch` tables: weight -> entry of candidate j, tagged with the
// pass so that they never need clearing) and from the current weights elsewhere.  When the last evaluating workgroup of pass P
// reports, all that is left is: reduce the partial sums, run the accept tests in iteration order, name the outcome, publish its
// descriptor and raise the flag - a few microseconds between two passes instead of the step's 15-20, and no pass is ever evaluated
// in vain.  The accepted candidate is committed to the float64 weights and the global weight image AFTER the flag: the next
// pass's evaluating workgroups, which copy that image meanwhile, apply the accepted entries to their LDS copies themselves
// (descriptor: pad[1] = their number, pad[2] bits 8.. = where their values are) - writing the same values twice is harmless.
// Every decision is made in iteration order on sums computed from the true current state: the sequential chain.
// ------------------------------------------------------------------------------------------------
constexpr int kSpecOutcomes = kMaxCand + 1;
// The sums of a pass under this schedule travel as PAIRS OF 64-BIT WORDS that carry the pass tag: {high half of the double, tag},
// {low half, tag}.  A 64-bit store arrives whole, so a pair whose two tags match is the value of that pass - and the evaluating workgroups
// need not wait for their stores to be acknowledged before they report, nor report at all: the step looks at the words themselves, in the
// loads it adds up anyway (two round trips through memory less between two passes).  Tags are the touch tables' pass tags
// (ChainParams.spec_gen + P + 1: never repeated, the host clears the words with the tables).
constexpr int kSpecPartSlots = 256;       // workgroups a record block is laid out for (one per compute unit at most); spec_rounds reads
                                          // them as 4 x 64 lanes with no remainder loop: chain_prepare keeps launches with more evaluating
                                          // workgroups off this schedule
static_assert(kSpecPartSlots == 4 * 64, "spec_rounds adds the workgroups' sums up as four loads per lane of one wave");
__device__ __forceinline__ size_t spec_part_index(int par, int j, int v, int b) {
    return ((((size_t)par * kMaxCand + j) * kPartialStride + v) * kSpecPartSlots + b) * 2;
}
__device__ __forceinline__ void spec_part_store(unsigned long long* rec, double s, unsigned tag) {
    const unsigned long long hi = (unsigned long long)(unsigned)__double2hiint(s), lo = (unsigned long long)(unsigned)__double2loint(s);
    __hip_atomic_store(rec, (hi << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 1, (lo << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kSpecRows = 2 * kMaxCand;            // rows (iterations t0 .. t0+2D-1) a round looks at: the candidates in flight and those of every outcome
struct SpecDesc {               // a prepared pass as the evaluating workgroups read it: sixteen words, so that ONE wave-wide load fetches
    PassDesc d;                 // the four outcomes' (pad[0] terminal, pad[1] accepted entries to apply first, pad[2] = own patch-value
    int tag;                    // slot | (slot of the accepted candidate's values + 1) << 8).  tag: the pass it describes (P + 1 >= 1),
    int pad_[7];                // stored after the other words have arrived: a record whose tag matches is whole
};
struct SpecState {
    double cand_lp[2][kSpecOutcomes][kMaxCand];   // log priors of the prepared candidates, by parity of their pass and outcome
    SpecDesc desc[2][kSpecOutcomes];              // their descriptors, written while the pass before is still being evaluated: after the
                                                  // decision the flag alone (ChainDev.prepared = (P + 1) << 2 | outcome) names one of them
    int ovf[2][kSpecOutcomes];                    // some candidate of that outcome leaves the fp16 range
    int o_cur;                                    // outcome slot the candidates of the pass in flight came from
    int rounds;                                   // diagnostics: rounds run, and wall-clock ticks (100 MHz) spent per phase:
    unsigned long long ticks[6];                  // touch tables | candidates | descriptors | wait for the pass | decide + publish | commit
};

// Evaluating workgroup of pass `launch` under the decision-between-passes schedule: wait for the flag, which names the outcome, and hand
// the workgroup that outcome's descriptor (lds_words[0..7]; lds_words[8]: the wait ended well).  Wave 0 waits, and every look of it asks
// for the flag AND for the four prepared descriptors (one wave-wide load): the look that sees the flag has the descriptor with it - no
// second round trip between the flag and the pass.  (A record read before the step had written it shows in its tag and is fetched again.)
__device__ __forceinline__ bool sync_eval_enter_spec(ChainDev* st, const SpecState* S, const PassDesc* pass, int launch, int par, int* lds_words) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int ok = 1, w = 0;
        if (launch == 0) {
            // the batch's first pass was prepared by the first step kernel, which was through before this launch began: its plain block
            w = __hip_atomic_load(reinterpret_cast<const int*>(pass + par) + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const int* const rec = reinterpret_cast<const int*>(&S->desc[par][0]);       // four records of sixteen words
            const unsigned long long t_begin = wall_clock64();
            int flag;
            for (;;) {
                flag = __hip_atomic_load(&st->prepared, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w = __hip_atomic_load(rec + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag = __builtin_amdgcn_readfirstlane(flag);
                if ((flag >> 2) >= launch) break;
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&st->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = 0; break; }
                if (wall_clock64() - t_begin > kSyncTimeoutTicks) {
                    if (lane == 0) __hip_atomic_store(&st->aborted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (ok) {
                const int o = flag & 3;
                if (__builtin_amdgcn_readlane(w, o * 16 + 8) != launch)      // (asked for before the record was there: now it is)
                    w = __hip_atomic_load(rec + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w = __builtin_amdgcn_ds_bpermute((o * 16 + (lane & 15)) << 2, w);
            }
        }
        if (lane < 8) lds_words[lane] = w;
        if (lane == 0) lds_words[8] = ok;
    }
    __syncthreads();
    return lds_words[8] != 0;
}

#ifndef NPBNN_SPEC_INLINE
#define NPBNN_SPEC_INLINE __attribute__((noinline))
#endif
struct SpecShared {            // LDS scratch of spec_rounds (beside StepShared)
    double red[kSpecOutcomes][kMaxCand][16];
    double cand_lp[kSpecOutcomes][kMaxCand];   // log priors of the candidates being prepared, per outcome
    double sel_lp[kMaxCand];                   // log priors of the candidates of the pass in flight
    double cur_ll, cur_lp;                     // log-likelihood / log prior of the current state
    double out_ll[kMaxCand];
    PassDesc desc[kSpecOutcomes];
    PassDesc cur;                              // descriptor of the pass in flight
    int ovf[kSpecOutcomes];
    int out_a[kMaxCand];
    int outcome, accepted, n_done, o_cur, go, pad_;
};

// The rounds of the step workgroup, P0 .. P_end-1 (ends early at the batch's last pass, or when a wait times out).
// Written for memory-level parallelism: a workgroup alone on its compute unit pays 1-2 us for every dependent global access, so each
// stage issues ALL its loads unconditionally (indices clamped to something valid, results selected afterwards) before the first
// use, what one round leaves for the next stays in LDS, and the parameter block is copied into registers once.
template <bool HAS_MASK>
__device__ NPBNN_SPEC_INLINE void spec_rounds(const ChainParams& c, int P0, int P_end, int n_eval_wgs, StepShared& sh, SpecShared& sp, int* lds_flag) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    // The parameter block sits in memory and the compiler must assume that any store may change it: every use of c.<field> below a
    // store would be a load of its own, in front of the load it feeds.  Everything the rounds need is copied out once, into
    // scalar registers (all of it is uniform).
    auto sg = [](auto v) { return __builtin_amdgcn_readfirstlane(v); };
    auto sgp = [](auto* ptr) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<decltype(ptr)>(((unsigned long long)hi << 32) | lo);
    };
    auto sgd = [](double v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
        return __hiloint2double((int)hi, (int)lo);
    };
    ChainDev* const st = sgp(c.st);
    SpecState* const S = sgp(c.spec);
    PassDesc* const pass = sgp(c.pass);
    const int lik_kind = sg(c.net.lik_kind), k_targets = sg(c.net.k_targets);
    const int M = sg(c.M), D = sg(c.D), K = sg(c.K), n_blocks = sg(c.n_blocks), nws = sg(c.n_weights_spec);
    const int prior_kind = sg(c.prior_kind);
    const double w_bound = sgd(c.w_bound);
    const int* const g_idx = sgp(c.idx);
    const double* const g_delta = sgp(c.delta);
    const int* const g_pos = sgp(c.pos);
    const int* const g_cnt = sgp(c.cnt);
    const double* const g_logu = sgp(c.log_u);
    const double* const g_hast = sgp(c.hastings);
    const unsigned long long* const g_spec_part = sgp(c.spec_part);
    // touch tables: one 16-byte record per weight and candidate - {pass tag, -, value} - so that a look-up is ONE gather
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4* const g_touch = reinterpret_cast<u32x4*>(sgp(c.spec_touch));
    double* const g_spec_pv = sgp(c.spec_pv);
    double* const Wc = sgp(c.w_cur);
    float* const g_image = sgp(c.image);
    const int l0_rows_img = sg(c.net.l0_rows);
    const double* const g_smult = sgp(c.sigma_mult);
    unsigned char* const g_out_acc = sgp(c.out_acc);
    double* const g_out_ll = sgp(c.out_ll);
    double* const g_out_lp = sgp(c.out_lp);
    int* const g_overflow = sgp(c.overflow);
    const long long n_rows = c.n_rows;
    const double lik_temp = sgd(c.lik_temp);
    const int sigma_given = sg(c.sigma_given), skip_round = sg(c.sync_test_skip), gen0 = sg(c.spec_gen);
    const bool has_sc = c.pscale != nullptr;
    constexpr bool has_mask = HAS_MASK;
    // arrays that may be absent read from something valid instead (the value is replaced by a constant afterwards)
    const float* const pscale_p = has_sc ? sgp(c.pscale) : reinterpret_cast<const float*>(g_pos);
    const double* const mask_p = has_mask ? sgp(c.mask) : Wc;
    // the priors this schedule runs with (the host checks): uniform, or normal with one scale per layer - 0.5 / scale^2 of a weight's
    // layer is selected by the layers' offsets, as chain_step does
    int woff[kMaxLayers];
    double l_his[kMaxLayers];
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        woff[l] = l < sg(c.net.n_layers) ? sg(c.net.L[l].w_off) : 0x7fffffff;
        l_his[l] = sgd(c.half_inv_s2[l]);
    }
    const size_t slot_stride = (size_t)kMaxCand * M;
    const int lim = sg(st->seg_end);
    const double temperature = sgd(st->temperature);
    const int nvals = partial_value_count(lik_kind, k_targets);
    // what the first round starts from: the descriptor, log priors and state the previous launch (or the batch's first step) left
    if (tid == 0) {
        const int q0 = P0 & 1;
        sp.cur = pass[q0];
        const int o0 = P0 == 0 ? 0 : S->o_cur;
        sp.o_cur = o0;
        for (int j = 0; j < kMaxCand; ++j) sp.sel_lp[j] = P0 == 0 ? st->cand_logPrior[0][j] : S->cand_lp[q0][o0][j];
        sp.cur_ll = st->logLik;
        sp.cur_lp = st->logPrior;
    }
    __syncthreads();
    unsigned long long tk = wall_clock64();
#define NPBNN_SPEC_TICK(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); S->ticks[k] += now_ - tk; tk = now_; } } while (0)

    for (int P = P0; P < P_end; ++P) {
        const int q = P & 1, qn = q ^ 1;
        const int t0 = sg(sp.cur.t0), n_pend = sg(sp.cur.n_cand);
        if (n_pend == 0) return;
        const int o_P = sg(sp.o_cur);
        const int slotP = (P % 3) * kSpecOutcomes + o_P;      // (patch values rotate through three sets: the evaluation of pass P + 1 still
                                                              //  reads the accepted candidate's of pass P while round P + 1 writes pass P + 2's)
        const double* const pvP = g_spec_pv + (size_t)slotP * slot_stride;
        const int slotN = ((P + 1) % 3) * kSpecOutcomes;
        double* const pvN = g_spec_pv + (size_t)slotN * slot_stride;
        const unsigned gen = (unsigned)(gen0 + P + 1);         // pass tag of the touch tables (never reused: the host clears them first)

        // operands of the decision and of the descriptors, on their way while the candidates are prepared
        double d_logu[kMaxCand], d_h[kMaxCand];
        int d_cnt[kMaxCand], d_acc_cnt = 0;
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) { d_logu[j] = 0.0; d_h[j] = 0.0; d_cnt[j] = 0; }
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j) {
                const int t = t0 + (j < n_pend ? j : 0);
                d_logu[j] = g_logu[t];
                d_h[j] = g_hast ? g_hast[t] : 0.0;
            }
        }
        if (tid < kSpecOutcomes) {                             // thread o: entries per candidate of outcome o's pass
            const int o = tid;
            const int ts = o == 0 ? t0 + n_pend : t0 + o;
#pragma unroll
            for (int k = 0; k < kMaxCand; ++k) d_cnt[k] = g_cnt[ts + k < K ? ts + k : 0];
            d_acc_cnt = g_cnt[o == 0 ? 0 : t0 + o - 1];
        }

        double dlp[kSpecOutcomes][kMaxCand];
#pragma unroll
        for (int o = 0; o < kSpecOutcomes; ++o)
#pragma unroll
            for (int k = 0; k < kMaxCand; ++k) dlp[o][k] = 0.0;
        if (tid < kSpecOutcomes) sp.ovf[tid] = 0;

        // ---- A. touch tables of the candidates in flight: weight -> the value candidate j gives it, tagged with this pass.  ALL of a
        //      candidate's entries go in before anything is looked up (a proposal may be wider than the workgroup) ----
        for (int e0 = 0; e0 < M; e0 += nthr) {
            const int e = e0 + tid;
            const bool ev = e < M;
            int ti[kMaxCand];
            double tv[kMaxCand];
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j) {
                const bool v = ev && j < n_pend;
                ti[j] = g_idx[v ? (size_t)(t0 + j) * M + e : 0];
                tv[j] = pvP[(size_t)j * M + (ev ? e : 0)];
                if (!v) ti[j] = -1;
            }
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j)
                if (ti[j] >= 0) {
                    u32x4 rec;
                    rec[0] = gen;
                    rec[1] = 0u;
                    rec[2] = (unsigned)__double2loint(tv[j]);
                    rec[3] = (unsigned)__double2hiint(tv[j]);
                    g_touch[(size_t)j * nws + ti[j]] = rec;
                }
        }
        __syncthreads();
        NPBNN_SPEC_TICK(0);
        for (int e0 = 0; e0 < M; e0 += nthr) {            // (one trip unless a proposal is wider than the workgroup)
            const int e = e0 + tid;
            const bool ev = e < M;
            // ---- stage 1: the pre-drawn entries of rows t0+1 .. (row r = iteration t0 + r; entries past a row's count are -1, as
            //      the pre-draw leaves them) ----
            int ri[kSpecRows], rpos[kSpecRows];
            double rd[kSpecRows];
            float rsc[kSpecRows];
            ri[0] = -1; rpos[0] = 0; rd[0] = 0.0; rsc[0] = 1.0f;
#pragma unroll
            for (int r = 1; r < kSpecRows; ++r) {
                const int t = t0 + r;
                const bool rowv = ev && r < n_pend + D && t < K && (r < n_pend || t < lim);
                const size_t k = rowv ? (size_t)t * M + e : 0;
                ri[r] = g_idx[k];
                rd[r] = g_delta[k];
                rpos[r] = g_pos[k];
                rsc[r] = pscale_p[k];
                if (!rowv) ri[r] = -1;
            }
#pragma unroll
            for (int r = 1; r < kSpecRows; ++r)
                if (!has_sc) rsc[r] = 1.0f;
            // ---- stage 2: what the entries of rows 1.. touch; candidate j accepted = outcome j + 1, which takes rows j+1 .. j+D: its
            //      k-th candidate's entry looks itself up in candidate j's table ----
            double rb[kSpecRows], rm[kSpecRows], rsw[kSpecRows];
            rb[0] = 0.0; rm[0] = 1.0; rsw[0] = 0.0;
            u32x4 rt[kMaxCand][kMaxCand];
            double rp[kMaxCand][kMaxCand];
#pragma unroll
            for (int r = 1; r < kSpecRows; ++r) {
                const int ic = ri[r] >= 0 ? ri[r] : 0;
                rb[r] = Wc[ic];
                rm[r] = has_mask ? mask_p[ic] : 1.0;
            }
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j)
#pragma unroll
                for (int k = 0; k < kMaxCand; ++k) {
                    const int r = j + 1 + k < kSpecRows ? j + 1 + k : 0;
                    const int ic = ri[r] >= 0 ? ri[r] : 0;
                    rt[j][k] = g_touch[(size_t)j * nws + ic];
                }
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j)
#pragma unroll
                for (int k = 0; k < kMaxCand; ++k) {
                    const int r = j + 1 + k < kSpecRows ? j + 1 + k : 0;
                    rp[j][k] = rt[j][k][0] == gen ? __hiloint2double((int)rt[j][k][3], (int)rt[j][k][2]) : rb[r];
                }
#pragma unroll
            for (int r = 1; r < kSpecRows; ++r) {
                double his = l_his[0];
#pragma unroll
                for (int l = 1; l < kMaxLayers; ++l) his = ri[r] >= woff[l] ? l_his[l] : his;
                rsw[r] = his;
            }
            // ---- stage 3: proposals and prior changes.  Outcome o >= 1 takes rows o .. o+D-1; outcome 0 (nothing accepted) shares the
            //      rows of outcome n_pend - every index below is a compile-time constant, so all of this stays in registers ----
#pragma unroll
            for (int o = 1; o <= kMaxCand; ++o) {
                if (o > n_pend) continue;
#pragma unroll
                for (int k = 0; k < kMaxCand; ++k) {
                    const int r = o + k;
                    if (k >= D || r >= kSpecRows) continue;
                    const int rr = r < kSpecRows ? r : 0;
                    const int i = ri[rr];
                    if (i < 0) continue;
                    const double his = rsw[rr];
                    const bool in16 = rpos[rr] < 0;
                    const double sc = (double)rsc[rr];
                    const double v = spec_entry<true>(w_bound, prior_kind, false, rp[o - 1][k], rd[rr], rm[rr], 1.0, his, 1.0, dlp[o][k]);
                    __hip_atomic_store(pvN + ((size_t)o * kMaxCand + k) * M + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (write-through: nothing to flush at the flag)
                    if (in16 && !(fabs(v * sc) <= (double)kF16Safe)) atomicOr(&sp.ovf[o], 1);
                    if (o == n_pend) {          // the same rows on the unchanged state: outcome 0
                        const double v0 = spec_entry<true>(w_bound, prior_kind, false, rb[rr], rd[rr], rm[rr], 1.0, his, 1.0, dlp[0][k]);
                        __hip_atomic_store(pvN + ((size_t)k) * M + e, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (in16 && !(fabs(v0 * sc) <= (double)kF16Safe)) atomicOr(&sp.ovf[0], 1);
                    }
                }
            }
        }
        if ((tid & ~63) < M) {                        // (a wave without entries adds exact zeros: no butterfly for it)
#pragma unroll
            for (int o = 0; o < kSpecOutcomes; ++o)
#pragma unroll
                for (int k = 0; k < kMaxCand; ++k) {
                    double v = dlp[o][k];
                    v = butterfly_sum_f64(v);
                    if ((tid & 63) == 0) sp.red[o][k][tid >> 6] = v;
                }
        } else if ((tid & 63) == 0) {
#pragma unroll
            for (int o = 0; o < kSpecOutcomes; ++o)
#pragma unroll
                for (int k = 0; k < kMaxCand; ++k) sp.red[o][k][tid >> 6] = 0.0;
        }
        __syncthreads();
        NPBNN_SPEC_TICK(1);
        // ---- descriptors of the prepared passes (thread o builds outcome o's; kept in LDS until one of them is published) ----
        if (tid < kSpecOutcomes) {
            const int o = tid;
            const int ts = o == 0 ? t0 + n_pend : t0 + o;
            int n_new = lim - ts;
            if (n_new > D) n_new = D;
            if (n_new < 0 || o > n_pend) n_new = 0;
            // log prior of the state outcome o leaves: unchanged, or the accepted candidate's
            const double base_lp = o == 0 ? sp.cur_lp : sp.sel_lp[o - 1];
            PassDesc d;
            d.t0 = ts;
            d.n_cand = n_new;
#pragma unroll
            for (int k = 0; k < kMaxCand; ++k) {
                double sj = 0.0;
                for (int w = 0; w < (nthr >> 6); ++w) sj += sp.red[o][k][w];
                sp.cand_lp[o][k] = base_lp + sj;
                d.cnt[k] = k < n_new ? d_cnt[k] : 0;
            }
            d.pad[0] = n_new == 0 ? 1 : 0;
            d.pad[1] = o == 0 ? 0 : d_acc_cnt;                                 // accepted entries the evaluation applies first
            d.pad[2] = (slotN + o) | (o == 0 ? 0 : ((slotP * kMaxCand + (o - 1) + 1) << 8));
            sp.desc[o] = d;
            // ... and in memory already, for the evaluating workgroups to pick from when the flag names the outcome: nothing but the flag is
            // stored between the decision and the next pass.  (Two round trips here, while the pass is still being evaluated.)
            SpecDesc* const pub = &S->desc[qn][o];
            int* const dst = reinterpret_cast<int*>(&pub->d);
            const int* const src = reinterpret_cast<const int*>(&d);
#pragma unroll
            for (int w = 0; w < 8; ++w) __hip_atomic_store(dst + w, src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(&pub->tag, P + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
        }

        // ---- C. pass P has been evaluated: decide it ----
        NPBNN_SPEC_TICK(2);
        if (skip_round == P + 1) return;
        {   // every wave waits for the sums it adds: it asks for a value's words of all workgroups (lanes: workgroups lane, lane + 64, ...)
            // and looks at their tags; complete, they are added in workgroup order - else asked for again.  Bounded like every wait here.
            const int lane = tid & 63, wave = tid >> 6, nwv = nthr >> 6;
            const unsigned long long* const part = g_spec_part;
            const unsigned long long t_begin = wall_clock64();
            bool ok = true;
            for (int item = wave; item < n_pend * nvals && ok; item += nwv) {
                const int j = item / nvals, v = partial_value_index(item % nvals, k_targets);
                const unsigned long long* src = part + spec_part_index(q, j, v, 0);
                for (;;) {
                    unsigned long long w0[4], w1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int b = lane + 64 * u < n_blocks ? lane + 64 * u : 0;
                        w0[u] = __hip_atomic_load(src + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w1[u] = __hip_atomic_load(src + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bool whole = true;
                    double s = 0.0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (lane + 64 * u < n_blocks) {
                            whole = whole && (unsigned)w0[u] == gen && (unsigned)w1[u] == gen;
                            s += __hiloint2double((int)(w0[u] >> 32), (int)(w1[u] >> 32));
                        }
                    if (__builtin_amdgcn_ballot_w64(!whole) == 0ull) {
                        s = butterfly_sum_f64(s);
                        if (lane == 0) sh.tot[j][v] = s;
                        break;
                    }
                    if (__hip_atomic_load(&st->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
                    if (wall_clock64() - t_begin > kSyncTimeoutTicks) {
                        __hip_atomic_store(&st->aborted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (tid == 0) *lds_flag = 1;
            __syncthreads();
            if (!ok) *lds_flag = 0;
        }
        __syncthreads();
        if (*lds_flag == 0) return;
        NPBNN_SPEC_TICK(3);
        __syncthreads();
        if (tid == 0) {
            int accepted = -1, n_done = n_pend;
            const double d_ll = sp.cur_ll, d_lp = sp.cur_lp;
#pragma unroll
            for (int j = 0; j < kMaxCand; ++j) {
                sp.out_a[j] = 0;
                if (j < n_pend && accepted < 0) {
                    const int t = t0 + j;
                    if (g_smult) {
                        double sgm[NPBNN_MAX_TARGETS];
                        for (int r = 0; r < k_targets; ++r) sgm[r] = st->sigma[r] * g_smult[(size_t)t * k_targets + r];
                        loglik_from_totals(sh.tot[j], lik_kind, k_targets, n_rows, lik_temp, 1, sgm, &sh.o);
                    } else {
                        loglik_from_totals(sh.tot[j], lik_kind, k_targets, n_rows, lik_temp, sigma_given, c.sigma_fixed, &sh.o);
                    }
                    const double lp = sp.sel_lp[j];
                    const double post_new = sh.o.loglik + lp, post_old = d_ll + d_lp;
                    const int a = ((post_new - post_old) * temperature + d_h[j] >= d_logu[j]) ? 1 : 0;
                    sp.out_a[j] = a;
                    sp.out_ll[j] = sh.o.loglik;
                    if (a) {
                        sp.cur_ll = sh.o.loglik;
                        sp.cur_lp = lp;
                        if (lik_kind == NPBNN_LIK_GAUSS)
                            for (int r = 0; r < k_targets; ++r) st->sigma[r] = sh.o.sigma[r];
                        accepted = j;
                        n_done = j + 1;
                    }
                }
            }
            const int o = accepted + 1;
            // ---- D. publish the pass that outcome selects: the flag names it; the book-keeping follows behind ----
            // (everything pass P + 1 reads was stored past the caches and has arrived - the patch values and the four descriptors during
            // the preparation, each thread waiting for its own stores ahead of a barrier: the flag needs no release fence, which would
            // write this XCD's whole L2 back, and nothing to wait for)
            __hip_atomic_store(&st->prepared, ((P + 1) << 2) | o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const PassDesc nx = sp.desc[o];
            {   // the plain descriptor block as well (a later launch of the batch starts from it)
                int* const dst = reinterpret_cast<int*>(pass + qn);
                const int* const src = reinterpret_cast<const int*>(&nx);
#pragma unroll
                for (int w = 0; w < 8; ++w) __hip_atomic_store(dst + w, src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sp.outcome = o;
            sp.accepted = accepted;
            sp.n_done = n_done;
            sp.go = nx.n_cand == 0 ? 0 : 1;
        }
        __syncthreads();
        NPBNN_SPEC_TICK(4);
        // ---- E. behind the flag: book-keeping, and the accepted candidate goes into the float64 weights and the global image ----
        const int acc = sp.accepted, o_sel = sp.outcome, n_done = sp.n_done;
        if (tid < kMaxCand && tid < n_done) {
            const int j = tid, t = t0 + j;
            g_out_acc[t] = (unsigned char)sp.out_a[j];
            g_out_ll[t] = sp.out_ll[j];
            g_out_lp[t] = sp.sel_lp[j];
        }
        if (tid >= 64 && tid < 64 + kSpecOutcomes * kMaxCand) {   // (kept in memory too: a later launch of the batch picks the chain up from there)
            const int o = (tid - 64) / kMaxCand, k = (tid - 64) % kMaxCand;
            S->cand_lp[qn][o][k] = sp.cand_lp[o][k];
        }
        if (acc >= 0) {
            const size_t row = (size_t)(t0 + acc) * M;
            for (int e = tid; e < M; e += nthr) {
                const int i = g_idx[row + e];
                const double v = pvP[(size_t)acc * M + e];
                const int pos = g_pos[row + e];
                const float sc = pscale_p[row + e];
                if (i >= 0) {
                    Wc[i] = v;
                    patch_image(g_image, pos, has_sc ? sc : 1.0f, v, l0_rows_img);
                }
            }
        }
        __syncthreads();                               // (sel_lp is read above and rewritten below)
        if (tid == 0) {
            if (acc >= 0) {
                st->logLik = sp.cur_ll;
                st->logPrior = sp.cur_lp;
                st->logPrior_rep = sp.cur_lp;
                st->n_accepted += 1;
            }
            st->t = t0 + n_done;
            st->n_passes += 1;
            S->o_cur = o_sel;
            if (sp.ovf[o_sel]) atomicOr(g_overflow, kFlagF16Range);
            S->rounds += 1;
            // the next round's pass in flight
            sp.cur = sp.desc[o_sel];
            sp.o_cur = o_sel;
            for (int k = 0; k < kMaxCand; ++k) sp.sel_lp[k] = sp.cand_lp[o_sel][k];
            // the committed entries sit in this XCD's L2; the evaluating workgroups of other XCDs read the image from the memory
            // side: write them back now, off the critical path (the NEXT pass applies this accept's entries itself)
            if (acc >= 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        __syncthreads();
        NPBNN_SPEC_TICK(5);
        if (sp.go == 0) return;
    }
#undef NPBNN_SPEC_TICK
}

#ifdef NPBNN_KERNELS_CHAIN
// two-stream schedule, first pair of launches of a round (both streams are idle then, so the two launches become eligible
// together): this one-wave kernel goes in front of the second launch on its stream and lets it through only when the step
// workgroup of the first has begun - the same guarantee sync_step_leave gives from then on
__global__ void sync_gate_kernel(ChainDev* st, int first_launch) {
    if (threadIdx.x == 0) (void)sync_wait_ge(st, &st->started, first_launch);
}
// exchange run on two streams: the first launch of the next swap interval sits on the stream that did NOT run the exchange
// kernels; this gate in front of it opens when exchange_apply_kernel has prepared its pass.  The wait spans an all-gather
// with the other ranks, so its bound is generous (2 s).
__global__ void sync_gate_exchanged_kernel(ChainDev* st, int n_exchanges) {
    if (threadIdx.x == 0) (void)sync_wait_ge(st, &st->exchanged, n_exchanges, 200000000ull);
}
#endif  // NPBNN_KERNELS_CHAIN

#ifdef NPBNN_KERNELS_CHAIN
// serial schedule: the step as a kernel of its own, between two evaluation kernels (and as the first launch of every batch)
__global__ void __launch_bounds__(1024) chain_step_kernel(const ChainParams* __restrict__ cp, int first_launch) {
    const ChainParams& c = *cp;           // device-resident parameter block; only the per-launch scalar travels as an argument
    __shared__ StepShared sh;
    if (!first_launch && c.pass[0].n_cand == 0) return;      // launched past the end of the batch
    if (first_launch) {
        if (c.spec) {
            int* z = reinterpret_cast<int*>(c.spec);
            for (int i = threadIdx.x; i < (int)(sizeof(SpecState) / sizeof(int)); i += blockDim.x) z[i] = 0;
        }
        __syncthreads();
    }
    StepPlan pl;
    pl.first = first_launch;
    pl.resum = first_launch;
    pl.dec = first_launch ? -1 : 0;
    pl.fly = -1;
    pl.out = 0;
    pl.launch = 0;
    chain_step<true>(c, pl, sh);
    if (first_launch) sync_step_leave(c.st, 0);       // (flag-ordered schedule: pass 0 is ready)
}
#endif  // NPBNN_KERNELS_CHAIN

// ------------------------------------------------------------------------------------------------
// exchange run: several chains (one per ctx: on this GPU and, through RCCL, on the other ranks') advance in segments of
// seg_len iterations and exchange temperatures between segments as MC3.run_mcmc does (BNN_mc3.py:94-112) - all of it enqueued
// on the chains' streams, no host round trip per segment.  After the last launch of segment s every chain writes its record
// [logPost, temperature, reached-the-end flag] (exchange_pack_kernel); the records are all-gathered in place; then every
// chain applies the same decision to its own temperature and prepares the first pass of the next segment
// (exchange_apply_kernel).  The swap proposal (j, k, log u) is pre-drawn by the host from the stream the reference's parent
// process draws from.  A chain that was given too few launches to finish a segment shows in the flags: every chain then
// poisons itself at that very exchange, nothing more is decided anywhere, and the host finishes that segment the slow way.
// ------------------------------------------------------------------------------------------------
constexpr int kRecDoubles = 4;          // record of a chain at an exchange: logPost, temperature, done flag, (spare)
struct ExchangeParams {
    double* rec;               // [n_seg][world * per_rank][kRecDoubles]; chain i sits at (i % world) * per_rank + i / world
    const int* swap_j;         // [n_seg] chain ids of the proposed swap
    const int* swap_k;
    const double* swap_logu;   // [n_seg]
    double* snap_w;            // [n_seg][n_weights] weights of this chain at the exchanges where it came out cold, or nullptr
    double* snap_state;        // [n_seg][4]: logLik, logPrior, temperature after the exchange, iterations done
    int world, per_rank, n_seg, seg_len;
    int my_slot;               // this chain's record index
    int n_weights;
};

#ifdef NPBNN_KERNELS_CHAIN
__global__ void exchange_pack_kernel(const ChainParams* __restrict__ cp, const ExchangeParams* __restrict__ xp, int s) {
    if (threadIdx.x != 0) return;
    const ChainDev* st = cp->st;
    double* r = xp->rec + ((size_t)s * xp->world * xp->per_rank + xp->my_slot) * kRecDoubles;
    r[0] = st->logLik + st->logPrior_rep;              // MCMC._logPost (BNN_env.py:497)
    r[1] = st->temperature;
    r[2] = (!st->poisoned && st->t >= st->seg_end) ? 1.0 : 0.0;
    r[3] = (double)st->t;
}

__global__ void __launch_bounds__(1024) exchange_apply_kernel(const ChainParams* __restrict__ cp, const ExchangeParams* __restrict__ xp, int s,
                                                              int next_launch, int overlapped) {
    const ChainParams& c = *cp;
    const ExchangeParams& x = *xp;
    __shared__ StepShared sh;
    __shared__ int s_go;
    ChainDev* st = c.st;
    const int n_rec = x.world * x.per_rank;
    if (threadIdx.x == 0) {
        const double* rec = x.rec + (size_t)s * n_rec * kRecDoubles;
        int all_done = st->poisoned ? 0 : 1;
        for (int i = 0; i < n_rec; ++i) all_done &= rec[(size_t)i * kRecDoubles + 2] == 1.0 ? 1 : 0;
        if (!all_done) {
            st->poisoned = 1;
        } else {
            // BNN_mc3.py:99-112: chains j, k swap temperatures when
            //   (logPost_k - logPost_j) * T_j + (logPost_j - logPost_k) * T_k >= log u
            const int j = x.swap_j[s], k = x.swap_k[s];
            const int rj = (j % x.world) * x.per_rank + j / x.world, rk = (k % x.world) * x.per_rank + k / x.world;
            const double pj = rec[(size_t)rj * kRecDoubles], tj = rec[(size_t)rj * kRecDoubles + 1];
            const double pk = rec[(size_t)rk * kRecDoubles], tk = rec[(size_t)rk * kRecDoubles + 1];
            const double r = (pk - pj) * tj + (pj - pk) * tk;
            if (j != k && r >= x.swap_logu[s]) {
                if (x.my_slot == rj) st->temperature = tk;
                else if (x.my_slot == rk) st->temperature = tj;
            }
            st->seg_idx = s + 1;
            if (s + 1 < x.n_seg) st->seg_end += x.seg_len;
        }
        if (x.snap_state) {
            double* q = x.snap_state + (size_t)s * NPBNN_XSTATE_DOUBLES;
            q[0] = st->logLik; q[1] = st->logPrior_rep; q[2] = st->temperature; q[3] = (double)st->t;
            for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) q[4 + j] = st->sigma[j];
        }
        s_go = all_done;
        // the launch after this kernel decides "the pass before it": there is none
        if (overlapped) { PassDesc& d = c.pass[(next_launch + 1) & 1]; d.n_cand = 0; d.t0 = st->t; }
    }
    __syncthreads();
    if (x.snap_w && s_go && st->temperature == 1.0) {        // the cold chain is the one the logger samples (BNN_mc3.py:118-122)
        double* dst = x.snap_w + (size_t)s * x.n_weights;
        for (int i = threadIdx.x; i < x.n_weights; i += blockDim.x) dst[i] = c.w_cur[i];
    }
    StepPlan pl;
    pl.first = 0;
    pl.resum = s_go;           // as the first launch of a batch does: batches of the segment-by-segment path start here
    pl.dec = -1;
    pl.fly = -1;
    pl.out = overlapped ? (next_launch & 1) : 0;
    pl.launch = next_launch;
    chain_step<true>(c, pl, sh);        // (a kernel of its own: the form that also keeps a weight-streamed chain's candidate image)
    sync_step_leave(c.st, next_launch);             // (flag-ordered schedule: the pass of launch next_launch is ready)
    if (threadIdx.x == 0) __hip_atomic_store(&st->exchanged, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
#endif  // NPBNN_KERNELS_CHAIN

}  // namespace npbnn
