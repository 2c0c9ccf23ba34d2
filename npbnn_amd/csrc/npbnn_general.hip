// The general device chain (npbnn_chain_run_general): Metropolis-Hastings iterations whose proposal is more than a list of
// perturbed weights - the reference's other proposal kernels (UpdateUniform, UpdateFixedNormal, UpdateNormalNormalized,
// np_bnn/BNN_mcmc.py:27-42,71-96), weight indicators (np_bnn/BNN_env.py:460,464) and feature indicators (:424-433).  Every
// iteration builds the FULL candidate on the device (weights, indicators, column override, log prior, Hastings term), packs its
// weight image, evaluates it with the single-candidate kernel and decides it - four small launches around one pass over X, no host
// round trip.  The random numbers are pre-drawn by the caller in the reference's order; nothing here draws.
#include "npbnn_ctx.hip.h"

namespace {

struct GenDev {                // chain state on the device
    double logLik, logPrior;
    double sigma[NPBNN_MAX_TARGETS];
    double cand_prior, cand_hastings;
    int n_accepted, pad_;
};

struct GenParams {
    GenDev* st;
    double *Wc, *Wp, *Weff;               // current / proposed / forward weights (layer 0 x indicators), packed
    double *indc, *indp;                  // weight indicators of layer 0 (or nullptr)
    double *findc, *findp, *colov;        // feature indicators and the column override they give (or nullptr)
    const double* feature_means;
    const double* mask;
    const int* idx; const double* val; const int* cnt;                      // [K][M] unique entries, [K]
    const int* h_idx; const double* h_val; const double* h_fac; const int* h_cnt;   // every draw (fixed-normal proposal)
    const int* layer_mask;                // [K]
    const int* ind_ptr; const int* ind_pos;
    const int* find_ptr; const int* find_pos; const int* find_use;          // find_use[t]: the override applies at iteration t
    const double* log_u; const double* hastings_in; const double* sigma_mult;
    unsigned char* out_acc; double* out_ll; double* out_lp;
    const double* partials;
    int n_weights, n0, F, M, K, kind, n_blocks;
    int prior_kind, has_ind_prior, sigma_given, pad2_;
    double prior_ind1, w_bound, temperature, lik_temp;
    double prior_scale[kMaxLayers];
    const double* prior_scale_w;
    double sigma_fixed[NPBNN_MAX_TARGETS];
    long long n_rows;
    NetMeta net;
};

// np.sum of a contiguous float64 array as numpy computes it (pairwise summation, numpy/_core/src/umath/loops_utils.h.src: blocks
// of up to 128 elements summed with eight accumulators, larger arrays split in halves rounded to a multiple of eight) - the
// normalising proposal divides by exactly that number
__device__ double numpy_block_sum(const double* a, int n) {        // n <= 128
    if (n < 8) {
        double r = -0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}
__device__ double numpy_sum(const double* a0, int n0) {             // the recursion, unrolled onto a small stack (depth <= 32)
    const double* pa[32];
    int pn[32], state[32];
    double left[32];
    int sp = 0;
    pa[0] = a0; pn[0] = n0; state[0] = 0; left[0] = 0.0;
    double ret = 0.0;
    while (sp >= 0) {
        const int n = pn[sp];
        if (n <= 128) {
            ret = numpy_block_sum(pa[sp], n);
            --sp;
            continue;
        }
        int n2 = n / 2;
        n2 -= n2 % 8;
        if (state[sp] == 0) {             // left half first
            state[sp] = 1;
            pa[sp + 1] = pa[sp]; pn[sp + 1] = n2; state[sp + 1] = 0;
            ++sp;
        } else if (state[sp] == 1) {      // the left half has returned
            left[sp] = ret;
            state[sp] = 2;
            pa[sp + 1] = pa[sp] + n2; pn[sp + 1] = n - n2; state[sp + 1] = 0;
            ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

__device__ __forceinline__ double block_sum_all(double v, double* red) {      // fixed order; every thread gets the result
    v = butterfly_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

__global__ void __launch_bounds__(1024) gen_propose_kernel(const GenParams* __restrict__ gp, int t) {
    const GenParams& g = *gp;
    __shared__ double red[16];
    __shared__ double lsum[kMaxLayers];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < g.n_weights; i += nt) g.Wp[i] = g.Wc[i];
    if (g.indc) for (int i = tid; i < g.n0; i += nt) g.indp[i] = g.indc[i];
    if (g.findc) for (int i = tid; i < g.F; i += nt) g.findp[i] = g.findc[i];
    __syncthreads();
    {   // the proposal's entries (unique positions: the caller kept the last draw of each)
        const size_t row = (size_t)t * g.M;
        const int n = g.cnt[t];
        for (int e = tid; e < n; e += nt) {
            const int i = g.idx[row + e];
            if (i < 0 || i >= g.n_weights) continue;
            double v = g.kind == NPBNN_PROP_FIXED_NORMAL ? g.val[row + e] : g.Wc[i] + g.val[row + e];
            if (g.kind != NPBNN_PROP_NORMAL_NORMALIZED) {            // (BNN_mcmc.py:78-82 divides by the sum instead of reflecting)
                if (v > g.w_bound) v = g.w_bound - (v - g.w_bound);
                if (v < -g.w_bound) v = -g.w_bound + (-g.w_bound - v);
            }
            g.Wp[i] = v;
        }
    }
    __syncthreads();
    if (g.kind == NPBNN_PROP_NORMAL_NORMALIZED) {
        const int lm = g.layer_mask[t];
        if (tid < g.net.n_layers) {
            const LayerMeta& L = g.net.L[tid];
            lsum[tid] = (lm >> tid) & 1 ? numpy_sum(g.Wp + L.w_off, L.out_dim * (L.in_dim + L.has_bias)) : 1.0;
        }
        __syncthreads();
        for (int l = 0; l < g.net.n_layers; ++l) {
            if (!((lm >> l) & 1)) continue;
            const LayerMeta& L = g.net.L[l];
            const int n = L.out_dim * (L.in_dim + L.has_bias);
            for (int i = tid; i < n; i += nt) g.Wp[L.w_off + i] = g.Wp[L.w_off + i] / lsum[l];
        }
        __syncthreads();
    }
    if (g.mask) {
        for (int i = tid; i < g.n_weights; i += nt) g.Wp[i] *= g.mask[i];
    }
    if (g.indc)        // UpdateBinomial: |ind - flips| (BNN_mcmc.py:98-99)
        for (int e = g.ind_ptr[t] + tid; e < g.ind_ptr[t + 1]; e += nt) {
            const int p = g.ind_pos[e];
            if (p >= 0 && p < g.n0) g.indp[p] = fabs(g.indc[p] - 1.0);
        }
    if (g.findc)
        for (int e = g.find_ptr[t] + tid; e < g.find_ptr[t + 1]; e += nt) {
            const int p = g.find_pos[e];
            if (p >= 0 && p < g.F) g.findp[p] = fabs(g.findc[p] - 1.0);
        }
    __syncthreads();
    // forward weights: layer 0 times its indicators (BNN_env.py:464 / RunPredictInd); column override from the feature indicators
    for (int i = tid; i < g.n_weights; i += nt) g.Weff[i] = (g.indc && i < g.n0) ? g.Wp[i] * g.indp[i] : g.Wp[i];
    if (g.findc)
        for (int f = tid; f < g.F; f += nt)
            g.colov[f] = (g.find_use[t] && g.findp[f] == 0.0) ? g.feature_means[f] : __longlong_as_double(0x7ff8000000000000ll);
    // log prior of the candidate in full (npBNN.calc_prior, BNN_env.py:180-194)
    double lp = 0.0;
    if (g.prior_kind != NPBNN_PRIOR_UNIFORM) {
        for (int l = 0; l < g.net.n_layers; ++l) {
            const LayerMeta& L = g.net.L[l];
            const int n = L.out_dim * (L.in_dim + L.has_bias);
            for (int i = tid; i < n; i += nt)
                lp += log_prior_density(g.prior_kind, g.Wp[L.w_off + i], g.prior_scale_w ? g.prior_scale_w[L.w_off + i] : g.prior_scale[l]);
        }
    }
    lp = block_sum_all(lp, red);
    double n_on = 0.0;
    if (g.has_ind_prior && g.indc)
        for (int i = tid; i < g.n0; i += nt) n_on += g.indp[i];
    n_on = block_sum_all(n_on, red);
    double h = 0.0;
    if (g.kind == NPBNN_PROP_FIXED_NORMAL) {      // sum over EVERY draw of logpdf(old) - logpdf(drawn) = (drawn^2 - old^2) / (2 d^2)
        const size_t row = (size_t)t * g.M;
        for (int e = tid; e < g.h_cnt[t]; e += nt) {
            const int i = g.h_idx[row + e];
            if (i < 0 || i >= g.n_weights) continue;
            const double o = g.Wc[i], d = g.h_val[row + e];
            h += (d * d - o * o) * g.h_fac[row + e];
        }
    }
    h = block_sum_all(h, red);
    if (tid == 0) {
        if (g.has_ind_prior && g.indc) lp += n_on * log(g.prior_ind1) + ((double)g.n0 - n_on) * log(1.0 - g.prior_ind1);
        g.st->cand_prior = lp;
        g.st->cand_hastings = h + (g.hastings_in ? g.hastings_in[t] : 0.0);
    }
}

__global__ void __launch_bounds__(1024) gen_decide_kernel(const GenParams* __restrict__ gp, int t) {
    const GenParams& g = *gp;
    __shared__ double tot[kPartialStride];
    __shared__ npbnn_eval_out o;
    __shared__ int s_acc;
    const int lik = g.net.lik_kind, k = g.net.k_targets;
    reduce_partials(g.partials, g.n_blocks, partial_value_count(lik, k), k, tot);
    if (threadIdx.x == 0) {
        GenDev* st = g.st;
        if (g.sigma_mult) {
            double sg[NPBNN_MAX_TARGETS];
            for (int q = 0; q < k; ++q) sg[q] = st->sigma[q] * g.sigma_mult[(size_t)t * k + q];
            loglik_from_totals(tot, lik, k, g.n_rows, g.lik_temp, 1, sg, &o);
        } else {
            loglik_from_totals(tot, lik, k, g.n_rows, g.lik_temp, g.sigma_given, g.sigma_fixed, &o);
        }
        const double lp = st->cand_prior;
        const int a = (((o.loglik + lp) - (st->logLik + st->logPrior)) * g.temperature + st->cand_hastings >= g.log_u[t]) ? 1 : 0;
        g.out_acc[t] = (unsigned char)a;
        g.out_ll[t] = o.loglik;
        g.out_lp[t] = lp;
        if (a) {
            st->logLik = o.loglik;
            st->logPrior = lp;
            st->n_accepted += 1;
            if (lik == NPBNN_LIK_GAUSS)
                for (int q = 0; q < k; ++q) st->sigma[q] = o.sigma[q];
        }
        s_acc = a;
    }
    __syncthreads();
    if (s_acc) {
        for (int i = threadIdx.x; i < g.n_weights; i += blockDim.x) g.Wc[i] = g.Wp[i];
        if (g.indc) for (int i = threadIdx.x; i < g.n0; i += blockDim.x) g.indc[i] = g.indp[i];
        if (g.findc) for (int i = threadIdx.x; i < g.F; i += blockDim.x) g.findc[i] = g.findp[i];
    }
}

struct DevBuf {                     // a device allocation freed at scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

template <typename T>
int upload(npbnn_ctx* ctx, DevBuf& b, const T* src, size_t n) {
    if (!src || n == 0) return NPBNN_OK;
    HIP_TRY(ctx, hipMalloc(&b.p, n * sizeof(T)));
    HIP_TRY(ctx, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return NPBNN_OK;
}

}  // namespace

extern "C" int npbnn_chain_run_general(npbnn_ctx* ctx, const npbnn_chain_cfg* cfg, const npbnn_general_cfg* gc, double* W_inout,
                                       const double* mask_packed, int32_t K, const double* log_u, uint8_t* out_accepted,
                                       double* out_loglik_prop, double* out_logprior_prop, npbnn_chain_result* result) {
    if (!ctx) return fail(nullptr, NPBNN_E_ARG, "null ctx");
    if (!cfg || !gc || !W_inout || K < 1 || !log_u || !out_accepted || !result) return fail(ctx, NPBNN_E_ARG, "chain_run_general: bad arguments");
    if (!ctx->arch_set) return fail(ctx, NPBNN_E_STATE, "chain_run_general: call npbnn_set_arch first");
    if (ctx->shard_n > 0) return fail(ctx, NPBNN_E_STATE, "chain_run_general: the context's rows are split over ranks (npbnn_set_row_shard): npbnn_chain_run only");
    if (gc->proposal_kind < NPBNN_PROP_NORMAL || gc->proposal_kind > NPBNN_PROP_NORMAL_NORMALIZED)
        return fail(ctx, NPBNN_E_ARG, "chain_run_general: proposal_kind=%d", gc->proposal_kind);
    if (gc->M < 1 || !gc->idx || !gc->val || !gc->cnt) return fail(ctx, NPBNN_E_ARG, "chain_run_general: the proposal's entry lists are missing");
    if (gc->proposal_kind == NPBNN_PROP_FIXED_NORMAL && (!gc->h_idx || !gc->h_val || !gc->h_fac || !gc->h_cnt))
        return fail(ctx, NPBNN_E_ARG, "chain_run_general: the fixed-normal proposal needs the list of every draw");
    if (gc->proposal_kind == NPBNN_PROP_NORMAL_NORMALIZED && !gc->layer_mask) return fail(ctx, NPBNN_E_ARG, "chain_run_general: layer_mask is missing");
    if (gc->ind_inout && (!gc->ind_ptr || !gc->ind_pos)) return fail(ctx, NPBNN_E_ARG, "chain_run_general: indicator flips are missing");
    if (gc->find_inout && (!gc->find_ptr || !gc->find_pos || !gc->find_use || !gc->feature_means))
        return fail(ctx, NPBNN_E_ARG, "chain_run_general: feature-indicator flips / means are missing");
    if (cfg->prior_kind < 0 || cfg->prior_kind > NPBNN_PRIOR_LAPLACE) return fail(ctx, NPBNN_E_ARG, "chain_run_general: prior_kind=%d", cfg->prior_kind);
    if (cfg->slope_idx) return fail(ctx, NPBNN_E_ARG, "chain_run_general: trainable activation slopes are not part of the general chain");
    const int lik = ctx->net.lik_kind;
    if (lik == NPBNN_LIK_NONE) return fail(ctx, NPBNN_E_STATE, "chain_run_general: the architecture has no likelihood");
    for (int t = 0; t < K; ++t)
        if (gc->cnt[t] < 0 || gc->cnt[t] > gc->M || (gc->h_cnt && (gc->h_cnt[t] < 0 || gc->h_cnt[t] > gc->M)))
            return fail(ctx, NPBNN_E_ARG, "chain_run_general: entry count of iteration %d outside 0..%d", t, gc->M);
    Dataset& d = ctx->ds[0];
    int rc = check_dataset_for_lik(ctx, d, lik);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nw = ctx->n_weights, F = ctx->arch.in_dim;
    const int n0 = ctx->net.L[0].out_dim * (ctx->net.L[0].in_dim + ctx->net.L[0].has_bias);
    const int k = ctx->net.k_targets;
    const size_t KM = (size_t)K * gc->M;
    for (int attempt = cfg->force_f32 ? 1 : 0; attempt < 2; ++attempt) {
        LaunchPlan lp;
        rc = plan_launch(ctx, 0, &lp, attempt, 1, false, true);
        if (rc) return rc;
        rc = ensure_work_buffers(ctx, lp.n_waves);
        if (rc) return rc;
        hipStream_t st = ctx->stream;
        DevBuf b_state, b_W, b_ind, b_find, b_means, b_mask, b_idx, b_val, b_cnt, b_hidx, b_hval, b_hfac, b_hcnt, b_lm, b_iptr, b_ipos, b_fptr, b_fpos,
            b_fuse, b_logu, b_hin, b_smult, b_out, b_psw, b_params, b_flags;
        GenParams g{};
        // state and work buffers
        HIP_TRY(ctx, hipMalloc(&b_state.p, sizeof(GenDev)));
        HIP_TRY(ctx, hipMalloc(&b_W.p, (size_t)3 * nw * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&b_flags.p, sizeof(int)));
        HIP_TRY(ctx, hipMemsetAsync(b_flags.p, 0, sizeof(int), st));
        GenDev init{};
        init.logLik = cfg->cur_loglik;
        init.logPrior = cfg->cur_logprior;
        for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) init.sigma[j] = cfg->cur_sigma[j];
        HIP_TRY(ctx, hipMemcpyAsync(b_state.p, &init, sizeof init, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(b_W.p, W_inout, (size_t)nw * sizeof(double), hipMemcpyHostToDevice, st));
        g.st = b_state.as<GenDev>();
        g.Wc = b_W.as<double>();
        g.Wp = g.Wc + nw;
        g.Weff = g.Wp + nw;
        if (gc->ind_inout) {
            HIP_TRY(ctx, hipMalloc(&b_ind.p, (size_t)2 * n0 * sizeof(double)));
            HIP_TRY(ctx, hipMemcpyAsync(b_ind.p, gc->ind_inout, (size_t)n0 * sizeof(double), hipMemcpyHostToDevice, st));
            g.indc = b_ind.as<double>();
            g.indp = g.indc + n0;
        }
        if (gc->find_inout) {
            HIP_TRY(ctx, hipMalloc(&b_find.p, (size_t)3 * F * sizeof(double)));
            HIP_TRY(ctx, hipMemcpyAsync(b_find.p, gc->find_inout, (size_t)F * sizeof(double), hipMemcpyHostToDevice, st));
            g.findc = b_find.as<double>();
            g.findp = g.findc + F;
            g.colov = g.findp + F;
            if ((rc = upload(ctx, b_means, gc->feature_means, (size_t)F))) return rc;
            g.feature_means = b_means.as<double>();
        }
        if ((rc = upload(ctx, b_mask, mask_packed, (size_t)nw))) return rc;
        g.mask = b_mask.as<double>();
        if ((rc = upload(ctx, b_idx, gc->idx, KM)) || (rc = upload(ctx, b_val, gc->val, KM)) || (rc = upload(ctx, b_cnt, gc->cnt, (size_t)K))) return rc;
        g.idx = b_idx.as<int>(); g.val = b_val.as<double>(); g.cnt = b_cnt.as<int>();
        if (gc->proposal_kind == NPBNN_PROP_FIXED_NORMAL) {
            if ((rc = upload(ctx, b_hidx, gc->h_idx, KM)) || (rc = upload(ctx, b_hval, gc->h_val, KM)) || (rc = upload(ctx, b_hfac, gc->h_fac, KM)) ||
                (rc = upload(ctx, b_hcnt, gc->h_cnt, (size_t)K)))
                return rc;
            g.h_idx = b_hidx.as<int>(); g.h_val = b_hval.as<double>(); g.h_fac = b_hfac.as<double>(); g.h_cnt = b_hcnt.as<int>();
        }
        if ((rc = upload(ctx, b_lm, gc->layer_mask, (size_t)K))) return rc;
        g.layer_mask = b_lm.as<int>();
        if (gc->ind_inout) {
            if ((rc = upload(ctx, b_iptr, gc->ind_ptr, (size_t)K + 1)) || (rc = upload(ctx, b_ipos, gc->ind_pos, (size_t)(gc->ind_ptr[K] > 0 ? gc->ind_ptr[K] : 1))))
                return rc;
            g.ind_ptr = b_iptr.as<int>(); g.ind_pos = b_ipos.as<int>();
        }
        if (gc->find_inout) {
            if ((rc = upload(ctx, b_fptr, gc->find_ptr, (size_t)K + 1)) ||
                (rc = upload(ctx, b_fpos, gc->find_pos, (size_t)(gc->find_ptr[K] > 0 ? gc->find_ptr[K] : 1))) || (rc = upload(ctx, b_fuse, gc->find_use, (size_t)K)))
                return rc;
            g.find_ptr = b_fptr.as<int>(); g.find_pos = b_fpos.as<int>(); g.find_use = b_fuse.as<int>();
        }
        if ((rc = upload(ctx, b_logu, log_u, (size_t)K))) return rc;
        g.log_u = b_logu.as<double>();
        if (cfg->sigma_mult || cfg->hastings) {
            if (!cfg->sigma_mult || !cfg->hastings || lik != NPBNN_LIK_GAUSS)
                return fail(ctx, NPBNN_E_ARG, "chain_run_general: sigma_mult and hastings go together, with the Gaussian likelihood");
            if ((rc = upload(ctx, b_smult, cfg->sigma_mult, (size_t)K * k)) || (rc = upload(ctx, b_hin, cfg->hastings, (size_t)K))) return rc;
            g.sigma_mult = b_smult.as<double>();
            g.hastings_in = b_hin.as<double>();
        }
        if (cfg->prior_scale_w && cfg->prior_kind != NPBNN_PRIOR_UNIFORM) {
            if ((rc = upload(ctx, b_psw, cfg->prior_scale_w, (size_t)nw))) return rc;
            g.prior_scale_w = b_psw.as<double>();
        }
        HIP_TRY(ctx, hipMalloc(&b_out.p, (size_t)K * (1 + 2 * sizeof(double)) + 64));
        g.out_ll = b_out.as<double>();
        g.out_lp = g.out_ll + K;
        g.out_acc = reinterpret_cast<unsigned char*>(g.out_lp + K);
        g.partials = ctx->d_partials;
        g.n_weights = nw; g.n0 = n0; g.F = F; g.M = gc->M; g.K = K; g.kind = gc->proposal_kind; g.n_blocks = lp.n_waves;
        g.prior_kind = cfg->prior_kind;
        g.has_ind_prior = gc->has_indicator_prior ? 1 : 0;
        g.prior_ind1 = gc->prior_ind1;
        g.sigma_given = cfg->sigma_given;
        g.w_bound = cfg->w_bound; g.temperature = cfg->temperature; g.lik_temp = cfg->lik_temp;
        for (int l = 0; l < kMaxLayers; ++l) g.prior_scale[l] = cfg->prior_scale[l];
        for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) g.sigma_fixed[j] = cfg->sigma[j];
        g.n_rows = d.n_rows;
        for (int l = 0; l < kMaxLayers; ++l)      // (fixed activation slopes: cfg->cur_slopes, as in npbnn_chain_run)
            ctx->net.act_prm[l] = (!cfg->slope_idx && l < cfg->n_slopes && l < NPBNN_MAX_LAYERS) ? (float)cfg->cur_slopes[l] : 0.f;
        g.net = ctx->net;
        HIP_TRY(ctx, hipMalloc(&b_params.p, sizeof(GenParams)));
        HIP_TRY(ctx, hipMemcpyAsync(b_params.p, &g, sizeof g, hipMemcpyHostToDevice, st));      // (pageable source: staged before the call returns)
        EvalParams p = make_params(ctx, d);
        p.partials = ctx->d_partials;
        p.inst_w = d.inst_w;
        p.use_classw = ctx->n_classw > 0 ? 1 : 0;
        rc = push_eval_params(ctx, p);
        if (rc) return rc;
        const GenParams* dgp = b_params.as<GenParams>();
        for (int t = 0; t < K; ++t) {
            hipLaunchKernelGGL(gen_propose_kernel, dim3(1), dim3(1024), 0, st, dgp, t);
            launch_pack_weights(ctx, g.Weff, g.colov, ctx->d_image, reinterpret_cast<int*>(b_flags.p));
            rc = launch_plain_eval(ctx, lp, 0);
            if (rc) return rc;
            hipLaunchKernelGGL(gen_decide_kernel, dim3(1), dim3(1024), 0, st, dgp, t);
        }
        HIP_TRY(ctx, hipGetLastError());
        std::vector<double> h_out((size_t)2 * K);
        std::vector<double> h_W((size_t)nw), h_ind((size_t)(gc->ind_inout ? n0 : 0)), h_find((size_t)(gc->find_inout ? F : 0));
        GenDev fin{};
        int flags = 0;
        HIP_TRY(ctx, hipMemcpyAsync(h_out.data(), g.out_ll, (size_t)2 * K * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_accepted, g.out_acc, (size_t)K, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(h_W.data(), g.Wc, (size_t)nw * sizeof(double), hipMemcpyDeviceToHost, st));
        if (gc->ind_inout) HIP_TRY(ctx, hipMemcpyAsync(h_ind.data(), g.indc, (size_t)n0 * sizeof(double), hipMemcpyDeviceToHost, st));
        if (gc->find_inout) HIP_TRY(ctx, hipMemcpyAsync(h_find.data(), g.findc, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&fin, b_state.p, sizeof fin, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&flags, b_flags.p, sizeof flags, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (flags & kFlagStructure) return fail(ctx, NPBNN_E_ARG, "chain_run_general: a layer-0 weight is not zero where the mask given to npbnn_set_layer_mask is");
        if (ctx->net.l0_f16 && (flags & kFlagF16Range)) {      // a candidate left the fp16 range: the whole batch again on the float32 path
            if (ctx->l0_option == NPBNN_L0_F16) return fail(ctx, NPBNN_E_RANGE, "chain_run_general: a layer-0 weight left the fp16 range during this batch");
            if (attempt == 0) continue;
        }
        memcpy(W_inout, h_W.data(), (size_t)nw * sizeof(double));
        if (gc->ind_inout) memcpy(gc->ind_inout, h_ind.data(), (size_t)n0 * sizeof(double));
        if (gc->find_inout) memcpy(gc->find_inout, h_find.data(), (size_t)F * sizeof(double));
        if (out_loglik_prop) memcpy(out_loglik_prop, h_out.data(), (size_t)K * sizeof(double));
        if (out_logprior_prop) memcpy(out_logprior_prop, h_out.data() + K, (size_t)K * sizeof(double));
        memset(result, 0, sizeof *result);
        result->loglik = fin.logLik;
        result->logprior = fin.n_accepted > 0 ? fin.logPrior : cfg->cur_logprior;
        for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) result->sigma[j] = fin.sigma[j];
        result->n_accepted = fin.n_accepted;
        result->n_passes = K;
        result->n_candidates = 1;
        result->schedule = NPBNN_SCHED_SERIAL;
        result->temperature = cfg->temperature;
        result->iterations_done = K;
        return NPBNN_OK;
    }
    return fail(ctx, NPBNN_E_RANGE, "chain_run_general: a layer-0 weight left the fp16 range on both paths");
}
