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
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) s += shfl_xor_f64(s, sh);
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
#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) gather_pos_kernel(const int* __restrict__ idx, long long n, const int* __restrict__ w2img,
                                                         const float* __restrict__ w2scale, int* __restrict__ pos, float* __restrict__ pscale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int w = idx[i];
    pos[i] = w >= 0 ? w2img[w] : kSkipPos;
    if (pscale) pscale[i] = (w >= 0 && w2scale) ? w2scale[w] : 1.0f;
}
#endif  // NPBNN_KERNELS_MAIN

__device__ __forceinline__ void patch_global_image(const ChainParams& c, int pos, float scale, double v) {
    if (pos == 0x7fffffff) return;                   // an entry the image does not hold (outside the layer-0 block structure: it is 0)
    if (pos < 0) {                                   // fp16-split layer-0 entry
        const float wv = (float)(v * (double)scale);
        _Float16 hi, lo;
        split_f16(wv, hi, lo);
        _Float16* img16 = reinterpret_cast<_Float16*>(c.image);
        const int h = pos & 0x7fffffff;
        img16[h] = hi;
        img16[h + 512] = lo;
    } else {
        c.image[pos] = (float)v;
    }
}

// block-wide sum of one double per thread, fixed order; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* red /*LDS, >= 16 doubles*/) {
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) v += shfl_xor_f64(v, sh);
    __syncthreads();                                  // `red` may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
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

__device__ __forceinline__ void chain_step(const ChainParams& c, const StepPlan pl, StepShared& sh) {
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
            const LayerMeta& L = c.net.L[l];
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
    if (pl.fly < 0) {   // serial schedule: whichever candidate wins, the next pass starts at t0+1 .. t0+n_pend: pull those rows of
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
    if (tid == 0 && n_pend > 0) {
        d_ll = st->logLik;
        d_lp = st->logPrior;
        d_temp = st->temperature;
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j)
            if (j < n_pend) {
                d_cand[j] = st->cand_logPrior[pl.dec][j];
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
                double s = 0.0;
                for (int b = lane; b < c.n_blocks; b += 64) s += src[b];
#pragma unroll
                for (int shf = 32; shf > 0; shf >>= 1) s += shfl_xor_f64(s, shf);
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
                        loglik_from_totals(sh.tot[j], lik_kind, c.net.k_targets, c.n_rows, c.lik_temp, c.sigma_given, c.sigma_fixed, &sh.o);
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
        if (a >= 0) {           // commit the accepted candidate: master weights and the global weight image
            const size_t row = (size_t)(t0 + a) * c.M;
            const int n = c.cnt[t0 + a];
            const double* pv = c.pv + (size_t)pl.dec * pv_stride;
            for (int e = tid; e < n; e += blockDim.x) {
                const int i = c.idx[row + e];
                if (i >= 0) {
                    const double v = pv[(size_t)a * c.M + e];
                    c.w_cur[i] = v;
                    patch_global_image(c, c.pos[row + e], c.pscale ? c.pscale[row + e] : 1.0f, v);
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
    {
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
                if (e < cn[j]) {
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
            for (int u = 0; u < ES; ++u) {
                const int i = ii[j][u];
                bb[j][u] = i >= 0 ? wcur[i] : 0.0;
                mm[j][u] = (i >= 0 && mask) ? mask[i] : 1.0;
                sw[j][u] = (i >= 0 && c.prior_scale_w) ? c.prior_scale_w[i] : 1.0;
            }
        auto make = [&](int j, int e, int i, double base, double d, double m, int pos, float sc, double scale_w) {
            double v = base + d;
            if (v > c.w_bound) v = c.w_bound - (v - c.w_bound);
            if (v < -c.w_bound) v = -c.w_bound + (-c.w_bound - v);
            v *= m;
            pv_out[(size_t)j * c.M + e] = v;
            if (pos < 0 && !(fabs(v * (double)sc) <= (double)kF16Safe)) atomicOr(c.overflow, kFlagF16Range);
            if (c.prior_kind != NPBNN_PRIOR_UNIFORM) {
                double his = half_inv_s2[0], lsc = layer_scale[0];     // (selected, not indexed: a register array indexed at run time
#pragma unroll                                                         //  goes to scratch)
                for (int q = 1; q < kMaxLayers; ++q) {
                    const bool past = i >= woff[q];
                    his = past ? half_inv_s2[q] : his;
                    lsc = past ? layer_scale[q] : lsc;
                }
                if (c.prior_scale_w) dlp[j] += prior_delta(c.prior_kind, v, base, scale_w);
                else if (c.prior_kind == NPBNN_PRIOR_NORMAL) dlp[j] -= (v * v - base * base) * his;
                else dlp[j] += prior_delta(c.prior_kind, v, base, lsc);
            }
        };
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
#pragma unroll
            for (int u = 0; u < ES; ++u)
                if (ii[j][u] >= 0) make(j, tid + u * (int)blockDim.x, ii[j][u], bb[j][u], dd[j][u], mm[j][u], pp[j][u], ss[j][u], sw[j][u]);
            if (cn[j] > ES * (int)blockDim.x) {
                const size_t row = (size_t)(t_new + j) * c.M;
                for (int e = tid + ES * blockDim.x; e < cn[j]; e += blockDim.x) {
                    const int i = c.idx[row + e];
                    if (i >= 0) make(j, e, i, wcur[i], c.delta[row + e], mask ? mask[i] : 1.0, c.pos[row + e], c.pscale ? c.pscale[row + e] : 1.0f,
                                     c.prior_scale_w ? c.prior_scale_w[i] : 1.0);
                }
            }
        }
    }
    NPBNN_STAMP(5);
#pragma unroll
    for (int j = 0; j < kMaxCand; ++j) {
#pragma unroll
        for (int shf = 32; shf > 0; shf >>= 1) dlp[j] += shfl_xor_f64(dlp[j], shf);
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
        if (early_prepared < launch || early_prepared == 0x7fffffff) ok = sync_wait_ge(st, &st->prepared, launch) ? 1 : 0;
        else if (__hip_atomic_load(&st->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
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

#ifdef NPBNN_KERNELS_MAIN
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
#endif  // NPBNN_KERNELS_MAIN

#ifdef NPBNN_KERNELS_MAIN
// serial schedule: the step as a kernel of its own, between two evaluation kernels (and as the first launch of every batch)
__global__ void __launch_bounds__(1024) chain_step_kernel(const ChainParams* __restrict__ cp, int first_launch) {
    const ChainParams& c = *cp;           // device-resident parameter block; only the per-launch scalar travels as an argument
    __shared__ StepShared sh;
    if (!first_launch && c.pass[0].n_cand == 0) return;      // launched past the end of the batch
    StepPlan pl;
    pl.first = first_launch;
    pl.resum = first_launch;
    pl.dec = first_launch ? -1 : 0;
    pl.fly = -1;
    pl.out = 0;
    pl.launch = 0;
    chain_step(c, pl, sh);
    if (first_launch) sync_step_leave(c.st, 0);       // (flag-ordered schedule: pass 0 is ready)
}
#endif  // NPBNN_KERNELS_MAIN

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

#ifdef NPBNN_KERNELS_MAIN
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
            double* q = x.snap_state + (size_t)s * 4;
            q[0] = st->logLik; q[1] = st->logPrior_rep; q[2] = st->temperature; q[3] = (double)st->t;
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
    chain_step(c, pl, sh);
    sync_step_leave(c.st, next_launch);             // (flag-ordered schedule: the pass of launch next_launch is ready)
    if (threadIdx.x == 0) __hip_atomic_store(&st->exchanged, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
#endif  // NPBNN_KERNELS_MAIN

}  // namespace npbnn
