/*
 * npbnn_hip.h — C ABI of the MI355X (gfx950) backend for npBNN's MCMC hot path.
 *
 * The reference (dsilvestro/npBNN, np_bnn 0.1.23) is pure Python and has no
 * FFI of its own; its boundary for this path is the Python call surface.  Each
 * entry point below names the reference function(s) it replaces (file:line
 * relative to the upstream repository root).  INTEGRATION.md shows the ctypes
 * stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative NPBNN_E_* code; the
 *     message is available from npbnn_last_error(ctx) (ctx may be NULL for
 *     errors raised before a context exists);
 *   - the caller owns every host buffer and may free it as soon as the call
 *     returns; the library owns all device memory for the lifetime of the ctx;
 *   - one ctx = one chain = one HIP stream on one device; a ctx is not
 *     thread-safe, distinct ctxs are independent;
 *   - host matrices are dense row-major; weights are float64 as in the
 *     reference and are converted to float32 on the device; the per-row
 *     arithmetic is float32, every cross-row sum is float64 with a fixed
 *     reduction order (results are run-to-run identical).
 */
#ifndef NPBNN_HIP_H
#define NPBNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPBNN_ABI_VERSION 1
#define NPBNN_MAX_LAYERS 8      /* weight matrices per network                    */
#define NPBNN_MAX_WIDTH 4096    /* max nodes of any hidden/output layer (layers of more than 128 nodes, and networks whose
                                   weights do not fit a compute unit's LDS, run on the weight-streamed path) */
#define NPBNN_MAX_TARGETS 16    /* max target columns for the Gaussian/count liks */
#define NPBNN_XSTATE_DOUBLES (4 + NPBNN_MAX_TARGETS)   /* doubles per exchange in npbnn_chain_job.out_state */

enum {
    NPBNN_OK = 0,
    NPBNN_E_ARG = -1,       /* bad argument / unsupported shape      */
    NPBNN_E_STATE = -2,     /* call order (e.g. eval before set_arch) */
    NPBNN_E_HIP = -3,       /* a HIP runtime call failed              */
    NPBNN_E_NOMEM = -4,
    NPBNN_E_COMM = -5,      /* RCCL failure                           */
    NPBNN_E_RANGE = -6,     /* value outside the fp16-split range      */
    NPBNN_E_SYNC = -7       /* a device-side wait of NPBNN_SCHED_OVERLAP2 timed out */
};

/* activation kinds — ActFun.activate selection, np_bnn/BNN_lib.py:50-87 */
enum { NPBNN_ACT_RELU = 0, NPBNN_ACT_LEAKY = 1, NPBNN_ACT_SWISH = 2, NPBNN_ACT_TANH = 3 };

/* output functions — np_bnn/BNN_lib.py:166-182 */
enum {
    NPBNN_OUT_SOFTMAX = 0,        /* SoftMax              :166-168 */
    NPBNN_OUT_IDENTITY = 1,       /* RegressTransform     :174-175 */
    NPBNN_OUT_SOFTPLUS_HALF = 2   /* RegressTransformError:177-182 */
};

/* likelihood kinds — np_bnn/BNN_lib.py:100-143 and np_bnn/BNN_lik.py:5-66 */
enum {
    NPBNN_LIK_CATEGORICAL = 0,     /* calc_likelihood                  BNN_lib.py:100-121 */
    NPBNN_LIK_GAUSS = 1,           /* calc_likelihood_regression       BNN_lib.py:123-131, sigma given or
                                      empirical (np.std of residuals, BNN_env.py:475-476) */
    NPBNN_LIK_GAUSS_PRED_SIGMA = 2,/* calc_likelihood_regression_error BNN_lib.py:134-143 */
    NPBNN_LIK_POISSON = 3,         /* poi_likelihood                   BNN_lik.py:5-14    */
    NPBNN_LIK_NEGBIN = 4,          /* negbin_likelihood                BNN_lik.py:16-30   */
    NPBNN_LIK_NEGBIN2D = 5,        /* negbin_likelihood2d              BNN_lik.py:33-49   */
    NPBNN_LIK_NEGBIN_BASE10 = 6,   /* negbin_likelihood_base10         BNN_lik.py:55-66   */
    NPBNN_LIK_NONE = 7             /* forward only                                         */
};

/* prior kinds — npBNN.__init__/calc_prior, np_bnn/BNN_env.py:135-150,180-194 */
enum { NPBNN_PRIOR_UNIFORM = 0, NPBNN_PRIOR_NORMAL = 1, NPBNN_PRIOR_CAUCHY = 2, NPBNN_PRIOR_LAPLACE = 3 };

enum { NPBNN_TRAIN = 0, NPBNN_TEST = 1 };

typedef struct npbnn_ctx npbnn_ctx;

/* Network description.  Layer l has a weight matrix out_dim[l] x (in_l + has_bias[l])
 * with the bias in COLUMN 0 (init_weight_prm, np_bnn/BNN_mcmc.py:9-25;
 * MatrixMultiplicationD, np_bnn/BNN_lib.py:154-162); in_0 = in_dim, in_l = out_dim[l-1].
 * Packed weights = the layer matrices concatenated, each row-major. */
typedef struct {
    int32_t n_layers;
    int32_t in_dim;
    int32_t out_dim[NPBNN_MAX_LAYERS];
    int32_t has_bias[NPBNN_MAX_LAYERS];
    int32_t act_kind;       /* hidden layers; the last layer has no activation (BNN_lib.py:253) */
    int32_t out_kind;
    int32_t lik_kind;
    int32_t n_targets;      /* k target columns (Gaussian / count likelihoods), else 0 */
    int32_t final_act;      /* 1: the activation also follows the last layer (RunHiddenLayer on its own, BNN_lib.py:190-191) */
} npbnn_arch;

/* Result of one evaluation.  sum_r / sum_r2 are the per-column residual moments
 * (targets - prediction) from which sigma, the Gaussian log-likelihood and the MSE
 * statistics (CalcAccuracyRegression / CalcLabelAccuracyRegression, BNN_lib.py:195-201)
 * all derive. */
typedef struct {
    double loglik;                       /* includes lik_temp */
    double sigma[NPBNN_MAX_TARGETS];     /* sigma actually used (given or empirical) */
    double sum_r[NPBNN_MAX_TARGETS];
    double sum_r2[NPBNN_MAX_TARGETS];
    int64_t n_rows;
} npbnn_eval_out;

/* ---- lifecycle ------------------------------------------------------------------ */
int npbnn_abi_version(void);
int npbnn_device_count(int* out);
int npbnn_create(int device_id, npbnn_ctx** out);
void npbnn_destroy(npbnn_ctx* ctx);
const char* npbnn_last_error(const npbnn_ctx* ctx);

/* ---- resident data: replaces npBNN._data/_labels/_test_data/_test_labels living in host
 * numpy arrays and being copied every iteration (tmp = bnn_obj._data + 0, BNN_env.py:388) ---- */
int npbnn_set_data_f64(npbnn_ctx* ctx, const double* X, int64_t n_rows, int32_t n_features, int which);
int npbnn_set_data_f32(npbnn_ctx* ctx, const float* X, int64_t n_rows, int32_t n_features, int which);
/* The chains of one run (MC3 replicates the model per chain, np_bnn/BNN_mc3.py:55-58) hold the same feature matrices: `ctx` uses
 * the device copies `owner` holds - training and test matrices, their fp16-split copies and scales - instead of uploading its
 * own (labels / targets / row weights stay per context: set them afterwards, then npbnn_set_arch).  Same device only.  The
 * owner's memory lives until the last borrower is destroyed or given data of its own; npbnn_set_data on an owner with
 * borrowers is an error. */
int npbnn_share_data(npbnn_ctx* ctx, npbnn_ctx* owner);

int npbnn_set_labels_i64(npbnn_ctx* ctx, const int64_t* y, int64_t n_rows, int which);
int npbnn_set_targets_f64(npbnn_ctx* ctx, const double* Y, int64_t n_rows, int32_t k, int which);
/* instance_weight / class_weight of calc_likelihood (BNN_lib.py:100-121); NULL clears. Train set only. */
int npbnn_set_row_weights(npbnn_ctx* ctx, const double* instance_w, int64_t n_rows,
                          const double* class_w, int32_t n_classes);
int npbnn_set_arch(npbnn_ctx* ctx, const npbnn_arch* arch);
/* Block structure of the first layer: `mask_packed` is the 0/1 mask of the network in packed-weight order (create_mask,
 * np_bnn/BNN_lib.py:16-47; npBNN.apply_mask, np_bnn/BNN_env.py:259-267; re-applied to every proposal, :461-462), or NULL for a
 * dense first layer.  Only its layer-0 part is looked at: blocks of 16 nodes x 16 features (32 on the fp16-split path) in which
 * the mask is all zero get no storage in the device's weight image and no matrix-core work - block_bnns.py's layouts (groups of
 * consecutive features wired to a few nodes each) shrink to their diagonal.  The caller promises that every weight it passes
 * from now on is zero where the mask is (the sampler's proposals are: BNN_env.py:462); a weight that is not fails the call that
 * brought it with NPBNN_E_ARG.  Results are the dense ones bit for bit.  After npbnn_set_arch; cleared by the next one. */
int npbnn_set_layer_mask(npbnn_ctx* ctx, const double* mask_packed);

/* ---- options / introspection.
 * NPBNN_OPT_L0_PRECISION selects how the first layer's product X.W0^T (MatrixMultiplicationD, BNN_lib.py:154-162; float64
 * np.dot in the reference) is computed:
 *   NPBNN_L0_F32   float32 matrix cores (v_mfma_f32_16x16x4_f32), bit-exact float32 fused-multiply-add chain;
 *   NPBNN_L0_F16   "fp16-split": every x and w is held as a pair of fp16 numbers (high + low part, ~22 significant bits,
 *                  features scaled per column by a power of two), three fp16 matrix-core products accumulated in float32;
 *                  same HBM bytes as float32, 5x less matrix-core time;
 *   NPBNN_L0_AUTO  fp16-split whenever the data and weights are representable (finite, in range), else float32; an
 *                  evaluation whose weights leave the fp16 range is transparently repeated in float32 (default). */
/* NPBNN_OPT_FAST_TAILS (default 1): launches that want nothing but the likelihood of a 2- or 3-layer network whose later layers
 * have <= 16 nodes (the layer loop of MCMC.mh_step, np_bnn/BNN_env.py:449-473, on every BASELINE shape) run on builds of the
 * evaluation kernel with the layer loop, the activation and the epilogue resolved at compile time; 0 keeps every launch on
 * the general build (same results bit for bit; for A/B timing).  NPBNN_INFO_FAST_TAILS: 1 when such launches would take them. */
/* NPBNN_OPT_PERSISTENT (default 1): npbnn_chain_run with cfg->schedule = NPBNN_SCHED_AUTO may pick NPBNN_SCHED_PERSIST for a chain
 * that has its GPU to itself; 0 keeps the automatic choice on kernel boundaries (NPBNN_SCHED_OVERLAP / _SERIAL). */
/* NPBNN_OPT_TRAINABLE_SLOPES (default 0): 1 reserves a slot per hidden layer in the weight image for the activation slope, so that the
 * candidates of a chain pass can each carry their own (npbnn_chain_cfg.slope_idx ...); such a network runs on the general builds. */
/* NPBNN_OPT_WIDE (default 0): the weight-streamed path.  A network runs on it BY ITSELF when a layer has more than 128 nodes or when its
 * weights would leave a compute unit's LDS fewer than 4 waves beside them (the reference's default n_nodes = [50, 5], np_bnn/BNN_env.py:20,
 * from ~700 features on, [32, 8] from ~980 (first layers of one or two output tiles leave it below eight waves); MatrixMultiplicationD, np_bnn/BNN_lib.py:154-162, takes any shape): every layer is then a
 * tiled matrix product whose operands both stream through LDS, weights from images resident in HBM (widths up to NPBNN_MAX_WIDTH, images
 * up to 2 GiB); a chain pass carries up to three candidates where the pass is one fused launch (narrow networks on many rows), else
 * one.  Same results as the resident path to rounding (not bit for bit: another order of float32 additions); the envelope and the
 * rates either side of the switch are in DESIGN.md 4.4.  1 = every network runs on it
 * (A/B timing, tests).  NPBNN_INFO_WIDE: 1 when the architecture set last runs on it.
 * NPBNN_INFO_F16_MOVED_COLUMNS / _F16_MAX_MOVE: the fp16-split copy scales every column by a power of two taken from its largest entry;
 * heavy-tailed columns (typical entries many powers of two below the largest) get that scale moved up so that the pair keeps their
 * typical entries (the largest entry then lies below 2^move instead of below 1; the weights' scale moves the other way): how many
 * columns of the training matrix were moved, and the largest move in powers of two (<= 12).  Valid once a launch has built the copy. */
enum { NPBNN_OPT_L0_PRECISION = 1, NPBNN_OPT_FAST_TAILS = 2, NPBNN_OPT_PERSISTENT = 3, NPBNN_OPT_TRAINABLE_SLOPES = 4, NPBNN_OPT_WIDE = 5 };
enum { NPBNN_L0_AUTO = 0, NPBNN_L0_F32 = 1, NPBNN_L0_F16 = 2 };
/* NPBNN_INFO_TURN_NS_OVERLAPPED / _BETWEEN: what NPBNN_SCHED_AUTO last measured for one launch turn (a pass, decided or void) of the
 * two persistent forms, in nanoseconds (0: never run on this context); NPBNN_INFO_IT_NS_OVERLAPPED / _BETWEEN: what an ITERATION of a
 * batch cost on each of them - the figure NPBNN_SCHED_AUTO compares, kept apart for batches of fewer than 256 iterations (reported
 * here when measured) and longer ones.  NPBNN_INFO_MAX_CANDIDATES: weight sets one pass over the
 * data can carry for this network (1-3: what fits a compute unit's LDS, and two from three layer-0 output tiles on) - the
 * candidates of a speculative chain pass, the chains of a group pass (npbnn_chains_run_batched), the sets of npbnn_predict_sets. */
enum { NPBNN_INFO_L0_F16 = 1, NPBNN_INFO_WAVES_PER_BLOCK = 2, NPBNN_INFO_N_CU = 3, NPBNN_INFO_FAST_TAILS = 4,
       NPBNN_INFO_TURN_NS_OVERLAPPED = 5, NPBNN_INFO_TURN_NS_BETWEEN = 6, NPBNN_INFO_MAX_CANDIDATES = 7,
       NPBNN_INFO_IT_NS_OVERLAPPED = 8, NPBNN_INFO_IT_NS_BETWEEN = 9, NPBNN_INFO_WIDE = 10,
       NPBNN_INFO_F16_MOVED_COLUMNS = 11, NPBNN_INFO_F16_MAX_MOVE = 12 };
int npbnn_set_option(npbnn_ctx* ctx, int option, int value);
int npbnn_get_info(npbnn_ctx* ctx, int what, int* out);

/* ---- one proposal evaluation: replaces the per-layer RunHiddenLayer loop + output function +
 * likelihood of MCMC.mh_step (BNN_env.py:449-473, 485-491) and of MCMC.__init__ (:299-319).
 *   W_packed     float64 packed weights (mask / indicators already applied by the caller, :461-464)
 *   act_prm      per-hidden-layer slope for NPBNN_ACT_LEAKY (ActFun._prm, BNN_lib.py:84-85) or NULL
 *   col_override length in_dim or NULL: entries that are not NaN replace that feature column by the
 *                given constant (data_transform_obj.transform, BNN_env.py:14-17)
 *   sigma        k values, or NULL = empirical sigma (BNN_env.py:475-476); Gaussian likelihood only
 *   confusion    n_classes x n_classes int64 [true label][argmax] or NULL; CalcAccuracy,
 *                CalcLabelAccuracy and CalcLabelFreq (BNN_lib.py:203-233) are functions of it */
int npbnn_eval(npbnn_ctx* ctx, const double* W_packed, const double* act_prm,
               const double* col_override, double lik_temp, const double* sigma, int which,
               npbnn_eval_out* out, int64_t* confusion);

/* ---- predictions: replaces RunPredict / RunPredictInd (BNN_lib.py:245-272).
 * out_y is n_rows x out_dim[last] float64; apply_out_fn=0 returns the last layer's pre-output values. */
int npbnn_predict(npbnn_ctx* ctx, const double* W_packed, const double* act_prm,
                  const double* col_override, int which, int apply_out_fn, double* out_y);

/* Posterior prediction: n_sets stored weight vectors (W_sets [n_sets][n_weights], act_prm_sets [n_sets][n_layers-1] or NULL)
 * against the resident matrix `which`; up to three sets share one streaming read of X.  out_y: [n_sets][n_rows][out_dim].
 * Replaces the loop over RunPredict of get_posterior_cat_prob (np_bnn/BNN_lib.py:375-381; also predictBNN :430, feature_importance
 * :504-597, get_posterior_est :715-748), which copies and re-reads the feature matrix once per posterior sample. */
int npbnn_predict_sets(npbnn_ctx* ctx, const double* W_sets, const double* act_prm_sets, int32_t n_sets, int which, int apply_out_fn,
                       double* out_y);

/* ---- timing hook for bench.py: launches the evaluation kernels `iters` times on the ctx stream
 * with weights already resident and returns the mean duration of the dominant kernel (HIP events
 * around each launch) and of the whole evaluation, in milliseconds. */
int npbnn_time_eval(npbnn_ctx* ctx, const double* W_packed, int iters, double* ms_main_kernel,
                    double* ms_total);

/* ---- device-resident Metropolis-Hastings iterations: replaces K consecutive calls of MCMC.mh_step
 * (np_bnn/BNN_env.py:381-532) on its default path - UpdateNormal proposals (np_bnn/BNN_mcmc.py:57-69), masks
 * (BNN_env.py:461-462), prior (npBNN.calc_prior, BNN_env.py:180-194), accept test (BNN_env.py:493-494) - with the
 * chain state resident on the GPU.  The host pre-draws the K iterations' random numbers from the very numpy Generator
 * stream the reference consumes (npbnn_host_predraw in libnpbnn_host.so) and passes them here:
 *   idx[t*M + j]   flat index into the packed weights of the j-th perturbed entry of iteration t, or -1 (skipped)
 *   delta[t*M + j] the normal deviate added to that entry;  cnt[t] entries are used in row t
 *   log_u[t]       log of the uniform draw of the accept test
 * W_inout holds the current weights on entry and the chain's weights after K iterations on return. */
typedef struct {
    int32_t prior_kind;                        /* NPBNN_PRIOR_* */
    double prior_scale[NPBNN_MAX_LAYERS];      /* npBNN._prior_scale, one per layer */
    double w_bound;                            /* reflection bound, INFINITY for none */
    double temperature;                        /* MCMC._temperature */
    double lik_temp;                           /* MCMC._lik_temp */
    int32_t sigma_given;                       /* Gaussian likelihood: 1 = use sigma[], 0 = empirical sigma */
    double sigma[NPBNN_MAX_TARGETS];           /* on entry: sigma for proposals when sigma_given */
    double cur_loglik, cur_logprior;           /* state of the chain on entry (MCMC._logLik / _logPrior) */
    double cur_sigma[NPBNN_MAX_TARGETS];       /* npBNN._error_prm on entry */
    int32_t force_f32;                         /* 1: run this batch on the float32 layer-0 path (after NPBNN_E_RANGE) */
    int32_t n_candidates;                      /* proposals evaluated per pass over X (speculative Metropolis-Hastings: iteration t and
                                                  t+1.. assuming the earlier ones are rejected; the chain is unchanged); 0 = as many
                                                  as fit (<= 3), 1 = strictly one evaluation per iteration */
    int32_t schedule;                          /* NPBNN_SCHED_AUTO / _SERIAL (evaluate a pass, decide it, evaluate the next) / _OVERLAP
                                                  (decide pass L-1 inside the launch that evaluates pass L, which was prepared assuming
                                                  pass L-1 rejects; a pass overtaken by an accept is dropped) / _OVERLAP2 (below).
                                                  The same chain whichever runs.  _AUTO: overlapped while fewer than ~37 % of the
                                                  iterations of the previous batch were accepted (75 % of the passes of 3) - as _PERSIST for a chain alone on
                                                  its GPU (NPBNN_OPT_PERSISTENT), else _OVERLAP - and _SERIAL above that; _OVERLAP2
                                                  only runs when asked for by name. */
    int32_t reserved_;
    /* regression with an estimated error parameter (BNN_env.py:435-442: every proposal multiplies sigma by pre-drawn factors,
     * multiplier_proposal_vector, BNN_mcmc.py:101-113): sigma_mult[t*n_targets + q] is the factor of target column q at iteration
     * t (1.0 = untouched), hastings[t] the proposal's Hastings term sum(log m).  The proposal of iteration t is evaluated with
     * sigma' = current sigma * factors (cur_sigma on entry; result->sigma on return), which becomes the chain's sigma when the
     * proposal is accepted.  NULL: sigma as sigma_given / sigma[] say. */
    const double* sigma_mult;
    const double* hastings;
    /* hyper-priors with one scale per input node or per weight (npBNN.sample_prior_scale, BNN_env.py:196-221: hyper_p = 2, 3):
     * the scale of every packed weight (n_weights values, layer matrices concatenated row-major like the weights), or NULL for
     * one scale per layer (prior_scale[] above).  Constant over the call: the Gibbs step that redraws them runs between calls. */
    const double* prior_scale_w;
    /* trainable activation slopes (ActFun(trainable=True), BNN_env.py:416-421,502-503; needs NPBNN_OPT_TRAINABLE_SLOPES): every
     * iteration first proposes new slopes from the accepted ones - UpdateNormal1D(acc_prm, d = 0.05, n = 1, bounds [0, 1]): entry
     * slope_idx[t] moves by slope_delta[t], reflected at 0 and 1 - evaluates its proposal with them, adds
     * log(r) * -sum(slopes') * r (r = 10) to the proposal's log prior and keeps them when the proposal is accepted.
     * cur_slopes: the accepted slopes on entry (result->slopes on return; n_slopes = hidden layers); slope_term_in_prior: 1 when
     * cur_logprior already holds that term for cur_slopes (it does after the chain's first accepted proposal: MCMC.__init__ computes
     * its prior without it, BNN_env.py:374).  slope_idx NULL: FIXED slopes (ActFun("genReLU", prm=...), BNN_lib.py:84-85) - the
     * n_slopes values of cur_slopes are the slopes of the hidden layers for every proposal of the batch (n_slopes = 0: none, slope 0). */
    const int32_t* slope_idx;
    const double* slope_delta;
    double cur_slopes[NPBNN_MAX_LAYERS];
    int32_t n_slopes;
    int32_t slope_term_in_prior;
} npbnn_chain_cfg;
#define NPBNN_SCHED_AUTO 0
#define NPBNN_SCHED_SERIAL 1
#define NPBNN_SCHED_OVERLAP 2
#define NPBNN_SCHED_OVERLAP2 3         /* the overlapped schedule with the launches alternating between two streams: the next launch's
                                         workgroups take the compute units over as the previous launch drains; what a kernel boundary
                                         guaranteed is guaranteed by device-side flags (release / acquire at agent scope).  A wait that
                                         times out ends the batch with NPBNN_E_SYNC (state untouched: retry with NPBNN_SCHED_OVERLAP).
                                         Opt-in: it counts on the two streams having hardware queues of their own and on nothing else
                                         sharing the GPU, neither of which HIP guarantees; a chain that shares its GPU runs _OVERLAP */

#define NPBNN_SCHED_PERSIST 4          /* the overlapped schedule as ONE persistent launch per batch round: every workgroup loops over
                                         the passes, a workgroup that has finished pass L goes straight on to pass L + 1 (no kernel
                                         boundary, no second stream, no launch gap), ordered by the same device-side flags as
                                         _OVERLAP2.  Needs every workgroup of the launch resident (grid <= compute units, the chain
                                         alone on its GPU); every wait is bounded: NPBNN_E_SYNC leaves the state untouched (retry
                                         with NPBNN_SCHED_OVERLAP; the library does so by itself and keeps the context off this schedule afterwards). */

#define NPBNN_SCHED_PERSIST_SERIAL 5   /* for chains that move: one persistent launch like _PERSIST, but a pass is DECIDED before the next one
                                         starts (evaluate pass L, decide it, evaluate pass L + 1 from the state it left - the order of the
                                         reference's loop, np_bnn/BNN_env.py:449-494), so no pass is ever evaluated from a state an accept
                                         has replaced.  The evaluating workgroups wait a few microseconds for the step workgroup between
                                         passes instead of a kernel boundary, a step kernel and a second boundary.  Same conditions,
                                         same bounded waits and the same fallback as _PERSIST.  _AUTO picks it over _PERSIST when the
                                         previous batch accepted more than ~7 % of its proposals. */

typedef struct {
    double loglik, logprior;                   /* state after the K iterations */
    double sigma[NPBNN_MAX_TARGETS];
    int64_t n_accepted;
    int32_t n_passes;                          /* passes over X used for the K iterations */
    int32_t n_candidates;                      /* candidates per pass actually used */
    int32_t n_void_passes;                     /* overlapped schedule: passes dropped because the pass before them accepted */
    int32_t schedule;                          /* schedule actually used (NPBNN_SCHED_SERIAL / _OVERLAP / _OVERLAP2) */
    double temperature;                        /* MCMC._temperature after the iterations (changes in an exchange run only) */
    int32_t iterations_done;                   /* K for npbnn_chain_run; an exchange run may stop a chain earlier (see below) */
    int32_t overflow;                          /* exchange run: 1 = the chain stopped before a proposal that leaves the fp16 range of the
                                                  layer-0 path (continue it with npbnn_chain_run, cfg->force_f32 = 1) */
    double slopes[NPBNN_MAX_LAYERS];           /* accepted activation slopes after the iterations (cfg->n_slopes of them) */
} npbnn_chain_result;

int npbnn_chain_run(npbnn_ctx* ctx, const npbnn_chain_cfg* cfg, double* W_inout, const double* mask_packed,
                    int32_t K, int32_t M, const int32_t* idx, const double* delta, const int32_t* cnt,
                    const double* log_u, uint8_t* out_accepted, double* out_loglik_prop, double* out_logprior_prop,
                    npbnn_chain_result* result);

/* ---- the general device chain: K iterations of MCMC.mh_step (np_bnn/BNN_env.py:381-532) for the sampler settings whose proposal is
 * more than a list of perturbed weights - the other proposal kernels (UpdateUniform np_bnn/BNN_mcmc.py:86-96: perturbations drawn from
 * numpy's global stream; UpdateFixedNormal :27-42: values replaced, Hastings term from every draw; UpdateNormalNormalized :71-82: the
 * perturbed layer divided by its sum), weight indicators (BNN_env.py:460,464: layer 0 multiplied by 0/1 indicators, flipped by
 * UpdateBinomial, BNN_mcmc.py:98-99, in the iterations that do not perturb layer 0; their Bernoulli prior, BNN_env.py:190-193) and
 * feature indicators (BNN_env.py:424-433: a switched-off feature reads as its mean, data_transform_obj :9-17).  Every iteration the
 * device builds the full candidate (weights, indicators, override, log prior in full, Hastings term), packs its weight image,
 * evaluates it (one pass over X) and decides it; nothing returns to the host in between.  All random numbers are pre-drawn by the
 * caller, in the reference's order, from the streams the reference uses:
 *   idx / val / cnt   per iteration the UNIQUE weights the proposal touches (the last draw of a position wins, as numpy's indexed
 *                     assignment does) and, per entry, the perturbation to add (NPBNN_PROP_NORMAL / _UNIFORM / _NORMAL_NORMALIZED) or
 *                     the value to install (NPBNN_PROP_FIXED_NORMAL)
 *   h_idx / h_val / h_fac / h_cnt   NPBNN_PROP_FIXED_NORMAL: EVERY draw - weight, drawn value, 0.5 / d^2 of its proposal width - for
 *                     the Hastings term sum(logpdf(old) - logpdf(drawn))
 *   layer_mask[t]     bit l: layer l was perturbed at iteration t (NPBNN_PROP_NORMAL_NORMALIZED renormalises those layers)
 *   ind_*             weight indicators: ind_inout the current 0/1 matrix of layer 0 (in / out), the flips of iteration t are
 *                     ind_pos[ind_ptr[t] .. ind_ptr[t+1])
 *   find_*            feature indicators likewise; find_use[t] = 1 where the override applies (iterations after adapt_stop)
 * cfg as for npbnn_chain_run (schedule, n_candidates, slopes unused: one candidate per pass, decided before the next). */
enum { NPBNN_PROP_NORMAL = 0, NPBNN_PROP_UNIFORM = 1, NPBNN_PROP_FIXED_NORMAL = 2, NPBNN_PROP_NORMAL_NORMALIZED = 3 };
typedef struct {
    int32_t proposal_kind;
    int32_t M;                                 /* capacity per iteration of the entry lists */
    const int32_t* idx;
    const double* val;
    const int32_t* cnt;
    const int32_t* h_idx;
    const double* h_val;
    const double* h_fac;
    const int32_t* h_cnt;
    const int32_t* layer_mask;
    double* ind_inout;                         /* or NULL: no weight indicators */
    const int32_t* ind_ptr;
    const int32_t* ind_pos;
    double prior_ind1;                         /* npBNN._prior_ind1 */
    int32_t has_indicator_prior;               /* npBNN._freq_indicator > 0 */
    int32_t reserved_;
    double* find_inout;                        /* or NULL: no feature indicators */
    const double* feature_means;
    const int32_t* find_ptr;
    const int32_t* find_pos;
    const int32_t* find_use;
} npbnn_general_cfg;
int npbnn_chain_run_general(npbnn_ctx* ctx, const npbnn_chain_cfg* cfg, const npbnn_general_cfg* gcfg, double* W_inout,
                            const double* mask_packed, int32_t K, const double* log_u, uint8_t* out_accepted,
                            double* out_loglik_prop, double* out_logprior_prop, npbnn_chain_result* result);

/* ---- stand-alone operators on host arrays (float64): the reference's call-surface helpers when user code calls them on an
 * explicit matrix instead of through the sampler.  Each call uploads, runs one device kernel and downloads.
 *   npbnn_op_activation  relu_f / leaky_relu_f / swish_f / tanh_f (BNN_lib.py:50-66); kind 4 = SoftPlus (:170-172); in place
 *   npbnn_op_output      SoftMax / RegressTransformError on the rows of a matrix (BNN_lib.py:166-182); in place; ind < 0: cols/2
 *   npbnn_op_likelihood  calc_likelihood* (BNN_lib.py:100-143) and the BNN_lik.py plug-ins on a prediction matrix
 *   npbnn_op_confusion   argmax predictions -> [true][predicted] counts and predicted-class counts
 *                        (CalcAccuracy / CalcLabelAccuracy / CalcLabelFreq, BNN_lib.py:203-233)
 *   npbnn_op_sse         per-column sum of squared errors, link 0 identity / 1 exp / 2 10^x
 *                        (CalcAccuracyRegression, CalcLabelAccuracyRegression BNN_lib.py:195-201; *_acc BNN_lik.py:81-99) */
int npbnn_op_activation(int device, int kind, double prm, double* inout, int64_t n);
int npbnn_op_output(int device, int out_kind, double* inout, int64_t rows, int32_t cols, int32_t ind);
int npbnn_op_likelihood(int device, int lik_kind, const double* pred, int64_t rows, int32_t cols, const int64_t* labels,
                        const double* targets, int32_t k, const double* inst_w, const double* class_w, int32_t n_class_w,
                        double lik_temp, const double* sigma, double* out);
int npbnn_op_confusion(int device, const double* pred, int64_t rows, int32_t cols, const int64_t* labels, int64_t* conf,
                       int64_t* pred_counts);
int npbnn_op_sse(int device, const double* pred, const double* targets, int64_t rows, int32_t cols_pred, int32_t k, int link,
                 double* out_per_col);

/* ---- MC3 temperature-swap exchange over RCCL (xGMI inside a node): replaces the multiprocessing pool
 * round trip of whole pickled chains in MC3.run_mcmc (np_bnn/BNN_mc3.py:94-112), of which the swap
 * decision only reads two scalars per chain.  One communicator per process / GPU; host buffers in and out.
 * Errors are reported through npbnn_last_error(NULL). */
typedef struct npbnn_comm npbnn_comm;
int npbnn_comm_unique_id(char out[128]);                                  /* rank 0; ship the bytes to every rank */
int npbnn_comm_init(int device_id, int rank, int nranks, const char id[128], npbnn_comm** out);
int npbnn_comm_allgather_f64(npbnn_comm* comm, const double* send, int count, double* recv /* nranks*count */);
int npbnn_comm_bcast_i64(npbnn_comm* comm, int64_t* buf, int count, int root);
void npbnn_comm_destroy(npbnn_comm* comm);
/* Which RCCL does this process run?  runtime_version: ncclGetVersion() of the library that is mapped (e.g. 22707), header_version:
 * NCCL_VERSION_CODE of the rccl.h this library was compiled against, path: the file the mapped library came from.
 * npbnn_comm_init refuses to start when the two differ in major.minor (a process that loaded another librccl.so.1 first - the one
 * bundled with a PyTorch wheel - would otherwise run this library's calls on it).  (No reference counterpart: BNN_mc3.py has no
 * communication library.) */
int npbnn_comm_runtime(int* runtime_version, int* header_version, char* path, int path_cap);
/* hipDeviceSynchronize on `device_id`: everything enqueued on that GPU by this process has completed (bench.py brackets its timed
 * region with it, next to a barrier of the communicator). */
int npbnn_device_synchronize(int device_id);

/* ---- one chain over several GPUs: the ROWS of the training matrix split over the processes of `comm` (rank r holds a contiguous
 * share as its ordinary training set), every process running the same chain from the same draws.  The reference evaluates the
 * likelihood of a proposal as one sum over all rows (np_bnn/BNN_lib.py:100-143 called from MCMC.mh_step, BNN_env.py:467-491); here a
 * pass leaves one record of partial sums per candidate on every rank (log-likelihood sum or residual moments), the records are
 * all-gathered - ncclAllGather on the chain's stream, kMaxCand x 33 doubles per rank - and every rank adds them in RANK ORDER and
 * takes the same decision: the chains stay bit-identical without another word between the ranks.  n_rows_total replaces the local
 * row count wherever the likelihood needs N (the empirical sigma of a regression).  After this call npbnn_chain_run runs on kernel
 * boundaries (NPBNN_SCHED_SERIAL: pass, gather, step) whatever schedule is asked for; npbnn_eval / npbnn_predict keep returning the
 * LOCAL rows' sums (the host adds what it needs across ranks: npbnn_amd/rowshard.py).  npbnn_chain_run_general, the batched and the
 * exchange runs refuse a sharded context.
 *   comm != NULL: the gather is enqueued on the stream (RCCL).  comm == NULL, gather != NULL: host-staged - the record comes to the
 *   host, gather(user, send, recv, count) must fill recv[n_ranks][count] (rank order) and return 0, the result goes back; the form
 *   the CPU-launched rehearsals use (processes that share one GPU cannot form an RCCL communicator).  n_ranks == 1: no gather.
 *   n_ranks == 0 switches sharding off. */
typedef int (*npbnn_gather_fn)(void* user, const double* send, double* recv, int32_t count);
int npbnn_set_row_shard(npbnn_ctx* ctx, npbnn_comm* comm, npbnn_gather_fn gather, void* user, int32_t rank, int32_t n_ranks,
                        int64_t n_rows_total);

/* ---- exchange run: the chains of an MC3 run (np_bnn/BNN_mc3.py:87-126) advance n_seg swap intervals of seg_len iterations with
 * the temperature swaps between them done on the GPU: replaces n_seg rounds of  pool.map(run_single_mcmc) -> read
 * [logPost, temperature] of every chain -> swap the temperatures of chains j, k when
 *   (logPost_k - logPost_j) * T_j + (logPost_j - logPost_k) * T_k >= log u           (BNN_mc3.py:99-112)
 * Everything is enqueued on the chains' streams: after the launches of a segment every chain writes its record, the records
 * are all-gathered in place (RCCL on the stream when `comm` spans several processes; chains of this process meet through
 * events), every chain applies the decision to its own temperature and starts the next segment.  No host round trip per
 * segment.  The swap proposals (swap_j, swap_k: chain ids; swap_logu) are pre-drawn by the caller from the stream the
 * reference's parent process draws from, identically on every rank.
 *   chain i lives on rank i % nranks as job i / nranks of that rank; n_chains = nranks * n_jobs; K = n_seg * seg_len rows per job.
 * The number of passes a segment needs is only known on the device; each segment is given launch_slack (1.25 if <= 0) times
 * the expected number.  If some chain has not finished a segment when it is exchanged (or stopped before an fp16 overflow), every
 * chain on every rank sees it in the records and stops at that exchange: *out_segments_done < n_seg, each job's
 * result->iterations_done says how far its chain got (its state and outputs are valid up to there), and the caller
 * finishes that segment with npbnn_chain_run and a host-side swap.
 * Failures between ranks: what can fail on one rank alone (its draws, its weights, its memory) is checked before anything is
 * enqueued, and the ranks agree on the outcome through one host-side all-gather of `comm` - on an error every rank returns one
 * (the failing rank its own, the others NPBNN_E_COMM) with nothing of the run in flight.  An error after that point (a launch or
 * a collective that fails in the middle of the run) aborts `comm` on the failing rank, so that its peers' pending collectives end
 * with an error instead of pairing with unrelated ones: the handle is dead afterwards (every call on it returns NPBNN_E_COMM);
 * destroy it and start again with a new one.
 *   out_records[s][i] = {logPost, temperature (before swap s), 1.0 if chain i had finished segment s, iterations done}
 *   job.out_state[s]  = {logLik, logPrior, temperature, iterations done, sigma[NPBNN_MAX_TARGETS]} of the chain after exchange s
 *                       (NPBNN_XSTATE_DOUBLES per exchange; sigma = the regression error parameter, BNN_env.py:500-501, which the
 *                       logger writes with the cold chain's row, :611-612)
 *   job.out_cold_w[s] = its weights at exchange s if it is the cold chain (temperature 1) afterwards (the logger's sample,
 *                       BNN_mc3.py:118-122); untouched otherwise */
typedef struct {
    npbnn_ctx* ctx;
    const npbnn_chain_cfg* cfg;
    double* W_inout;
    const double* mask_packed;
    int32_t M, chain_id;
    const int32_t* idx;
    const double* delta;
    const int32_t* cnt;
    const double* log_u;
    uint8_t* out_accepted;
    double* out_loglik_prop;                   /* or NULL */
    double* out_logprior_prop;                 /* or NULL */
    double* out_state;                         /* [n_seg][NPBNN_XSTATE_DOUBLES] or NULL */
    double* out_cold_w;                        /* [n_seg][n_weights] or NULL */
    npbnn_chain_result* result;
} npbnn_chain_job;
int npbnn_chains_run_exchange(npbnn_comm* comm /* NULL: the chains of this process only */, npbnn_chain_job* jobs, int32_t n_jobs,
                              int32_t n_chains, int32_t seg_len, int32_t n_seg, const int32_t* swap_j, const int32_t* swap_k,
                              const double* swap_logu, double launch_slack, double* out_records /* [n_seg][n_chains][4] */,
                              int32_t* out_segments_done);

/* ---- group pass: 2 or 3 chains of one model (the per-chain replicas MC3 makes, np_bnn/BNN_mc3.py:55-75: same data, same network)
 * that live on ONE GPU advance K iterations each, every launch evaluating one proposal PER CHAIN against a single streaming
 * read of the feature matrix: replaces n_jobs calls of MCMC.mh_step per iteration, each with its own copy of and pass over X
 * (BNN_mc3.py:80-85 through a pool of processes).  Candidate slot j of a launch belongs to chain j - its committed weight image
 * patched with its own pre-drawn perturbation - and its sums go to that chain's step, which runs in a workgroup of its own inside
 * the same launch while the next proposals are evaluated (the overlapped schedule of npbnn_chain_run, one candidate per chain:
 * a chain that accepts loses the one proposal that was in flight).  Every slot does useful work whatever the acceptance rate,
 * where a single chain's speculative candidates are wasted after its first accept of a pass.  Each chain is exactly the chain
 * npbnn_chain_run gives it (same decisions; its log-likelihood sums are formed from a different number of workgroup partials, so
 * they may differ in the last bits).  The jobs are filled in as for npbnn_chains_run_exchange (chain_id, out_state, out_cold_w
 * unused); the contexts must share their feature matrix (npbnn_share_data) and have the same architecture, mask and likelihood.
 * NPBNN_E_ARG when n_jobs weight images do not fit the LDS of a compute unit together; NPBNN_E_RANGE as npbnn_chain_run. */
int npbnn_chains_run_batched(npbnn_chain_job* jobs, int32_t n_jobs, int32_t K);

/* timing hook for a speculative chain pass: the evaluation kernel with n_candidates weight sets (0 = as many as fit),
 * `iters` back-to-back launches between one pair of HIP events on the ctx stream; mean milliseconds per launch */
int npbnn_time_pass(npbnn_ctx* ctx, const double* W_packed, int n_candidates, int iters, double* ms_kernel, int* used_candidates);

/* timing hook for the weight-streamed path (NPBNN_INFO_WIDE): `iters` back-to-back passes between one pair of HIP events - ms_pass:
 * all of a pass's launches (the layers' products, the likelihood kernel); ms_layer0: the first layer's product alone (with the
 * reduction of its K-slices, when it is cut into any) - the contraction over the feature matrix, the kernel the matrix cores'
 * roofline is quoted for.  info[0..3]: rows x outputs of a workgroup's block of that product, its K-slices, its workgroups.
 * NPBNN_E_STATE when the architecture set last runs on the LDS-resident path. */
int npbnn_time_wide(npbnn_ctx* ctx, const double* W_packed, int iters, double* ms_layer0, double* ms_pass, int* info);

/* page-locked host memory for the per-batch inputs of npbnn_chain_run (idx / delta / cnt / log_u): arrays drawn straight into
 * such memory are uploaded asynchronously at link speed instead of through the driver's bounce buffer.  (The reference has no
 * counterpart: its draws never leave the host, np_bnn/BNN_mcmc.py:57-69.) */
int npbnn_pinned_alloc(size_t bytes, void** out);
void npbnn_pinned_free(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* NPBNN_HIP_H */
