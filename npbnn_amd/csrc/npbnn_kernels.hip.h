// Device code of the npBNN hot path for gfx950 (MI355X, CDNA4).
//
// One evaluation = one streaming pass over the resident feature matrix X:
//   pack_weights_kernel : float64 packed weights -> float32 MFMA "fragment image" (+ padded biases); in the
//                         device-resident chain the image is instead patched entry by entry (chain_step_kernel)
//   eval_kernel<MT0>    : fused forward pass of the whole MLP + likelihood terms (+ confusion counts,
//                         + optional prediction output); per-wave float64 partial sums
//   finalize_kernel     : fixed-order float64 reduction of the partials -> log-likelihood, sigma, moments
//
// Mapping to the reference (np_bnn 0.1.23): the layer loop of MCMC.mh_step (BNN_env.py:449-473) /
// RunPredict (BNN_lib.py:245-256): RunHiddenLayer -> MatrixMultiplicationD (+ bias = column 0) -> ActFun.eval
// (BNN_lib.py:184-193,154-162,83-87); output functions (BNN_lib.py:166-182); likelihoods
// (BNN_lib.py:100-143, BNN_lik.py:5-66); accuracy reductions (BNN_lib.py:195-233).
//
// Design:
//  * a wavefront owns a 16-row tile of X and computes the TRANSPOSED problem  H^T = W . X^T  with
//    v_mfma_f32_16x16x4_f32: A = weight fragment (16 units x 4 k), B = X^T (4 k x 16 rows), so the
//    accumulator of layer l (unit on the register/lane-group index, data row on lane&15) is directly the
//    B operand of layer l+1 - the whole MLP chains through the matrix cores with no data movement
//    between layers; bias is the initial accumulator; activations are elementwise on accumulators.
//  * X tiles arrive by LDS-DMA (global_load_lds_dwordx4, one 1-KiB piece = 16 rows x 16 features per
//    wave-instruction) into a private per-wave ring; the wave that issued a piece waits for it with a
//    counted s_waitcnt vmcnt(N); no workgroup barrier in the main loop.
//  * all weights live in LDS as a lane-linear fragment image (ds_read_b128, conflict free), staged once
//    per persistent workgroup.
//  * softmax / log-likelihood: 4-lane shuffle reductions (the 4 lane groups of a data row), per-row terms
//    in float32, every cross-row sum in float64, one partial per wave, fixed order -> deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "npbnn_hip.h"

namespace npbnn {

constexpr int kMaxLayers = NPBNN_MAX_LAYERS;
constexpr int kMaxMT = NPBNN_MAX_WIDTH / 16;   // 16-unit tiles per layer
#ifndef NPBNN_RING
#define NPBNN_RING 4
#endif
constexpr int kRing = NPBNN_RING;              // X ring slots (1 KiB each) per wave; kRing-1 pieces stay in flight
constexpr int kMaxWavesPerBlock = 16;
constexpr int kAuxSlots = 4;                   // per-wave row-aux buffers (labels / weights / targets)
// per-wave aux slot: labels (64 B) + instance weights (64 B) + 16 x k targets; sized per network
__host__ __device__ inline int aux_bytes(int k_targets) { return 128 + 64 * k_targets; }
// likelihoods that combine several outputs of one row (predicted sigma, count data) exchange them through 1 KiB of LDS
__host__ __device__ inline bool lik_needs_row_scratch(int lik_kind) {
    return lik_kind >= NPBNN_LIK_GAUSS_PRED_SIGMA && lik_kind <= NPBNN_LIK_NEGBIN_BASE10;
}
__host__ __device__ inline int wave_lds_bytes(int k_targets, int lik_kind) {
    return kRing * 1024 + kAuxSlots * aux_bytes(k_targets) + (lik_needs_row_scratch(lik_kind) ? 1024 : 0);
}
constexpr int kPartialStride = 1 + 2 * NPBNN_MAX_TARGETS;   // loglik, sum_r[16], sum_r2[16]

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct LayerMeta {
    int kt;        // 16-wide k tiles of the input dimension
    int mt;        // 16-wide tiles of the output dimension
    int frag_off;  // float offset of the fragment block in the image
    int bias_off;  // float offset of the padded bias (16*mt floats)
    int in_dim, out_dim, has_bias;
    int w_off;     // double offset of the layer matrix in the packed weights
};

struct NetMeta {
    int n_layers;
    int image_floats;   // total floats of the image (multiple of 256)
    int classw_off;     // float offset of class weights (NPBNN_MAX_WIDTH floats)
    int act_kind, out_kind, lik_kind, n_out, k_targets;
    int final_act;      // apply the activation to the last layer's output too (RunHiddenLayer on its own)
    int l0_f16;         // layer 0 runs on the fp16-split path (x = xh + xl, w = wh + wl; 3 f16 MFMAs, f32 accumulate)
    LayerMeta L[kMaxLayers];
    float act_prm[kMaxLayers];
};

constexpr int kMaxCand = 3;    // candidates evaluated per pass over X by a speculative chain

// One pass of a device-resident chain evaluates, against a single streaming read of X, the proposal of iteration t0 and
// the proposals of iterations t0+1 .. t0+n_cand-1 *under the assumption that the earlier ones are rejected* (each is the
// current state plus its own pre-drawn perturbation).  The step kernel then decides them in order and stops at the first
// accepted one: the chain is exactly the sequential Metropolis-Hastings chain ("prefetching" / speculative MH).
struct PassDesc {
    int t0;                   // first iteration evaluated by the pass
    int n_cand;               // candidates in the pass (0: the batch is finished, the evaluation kernel exits at once)
    int cnt[kMaxCand];        // touched entries per candidate
    int pad[3];
};

struct ChainParams;

struct EvalParams {
    const float* X;           // [n_tiles*16][Fp] zero padded; in fp16-split mode the same bytes hold, per 8 features,
                              // 8 x fp16 high parts then 8 x fp16 low parts of the column-scaled values
    const int* labels;        // [n_tiles*16], -1 on padding rows
    const float* targets;     // [n_tiles*16][k] (k = k_targets), 0 on padding rows
    const float* inst_w;      // [n_tiles*16] or nullptr
    const float* image;       // float32 fragment image of the weights (global), DMA-copied into LDS
    double* partials;         // [2][candidate][kPartialStride][n_workgroups] (pass parity first)
    unsigned* confusion;      // [n_out*n_out] or nullptr
    float* y_out;             // [n_rows][n_out] or nullptr
    long long n_rows;
    int n_tiles;
    int Fp;
    int use_classw;
    int predict_mode;         // 0 none, 1 raw last-layer values, 2 output function applied
    int weight_sets;          // 0: the D candidates of a launch are patched copies of ONE image (chain pass); 1: D independent
                              // weight sets, image j at image + j*image_floats, predictions of set j at y_out + j*n_rows*n_out
                              // (posterior prediction: several stored samples per streaming read of X)
    // speculative multi-candidate pass of a device-resident chain (nullptr / unused for a plain evaluation):
    int has_pass;             // chain pass: pass_desc[parity] says which candidates this launch evaluates (parity 0 outside the
    int pad_pass_;            // overlapped schedule).  The descriptors live INSIDE this block - the step writes them here - so the
    PassDesc pass_desc[2];    // evaluation reads them with the rest of its parameters instead of through one more dependent load
    const double* pv;         // [2][kMaxCand][M] proposed values of the touched entries of each candidate
    const int* pos;           // [K][M] image position of every pre-drawn entry (w2img gather)
    const float* pscale;      // [K][M] fp16-split column scale of every pre-drawn entry, or nullptr
    int M;
    const ChainParams* chain;  // overlapped chain schedule: the last workgroup of the launch runs chain_step (else nullptr)
    unsigned long long* stamps;   // diagnostics only (NPBNN_EVAL_STAMPS=1 in npbnn_time_pass): [workgroup][8] wall-clock stamps, else nullptr
    NetMeta net;
};

// ------------------------------------------------------------------------------------------------
// pack: float64 packed weights -> fragment image
//   frag_l[((kt*MT + mt)*64 + lane)*4 + s] = W_l[o = 16mt + (lane&15)][c = 16kt + 4(lane>>4) + s]
//   (bias column excluded, zero outside the matrix); bias_l[o] = W_l[o][0] when the layer has a bias.
//   Layer 0 with a column override (data_transform_obj, BNN_env.py:14-17): an overridden feature column
//   is the constant v_c for every row, so its contribution v_c*W0[o][c] moves into the bias and the
//   fragment entry becomes 0 - no extra pass over X.
// ------------------------------------------------------------------------------------------------
// fp16 split of a float: hi = fp16(v), lo = fp16(v - hi); hi + lo carries ~22 significant bits of v
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

constexpr float kF16Safe = 60000.0f;   // |value| above this does not survive fp16 (max 65504)

// One item of the weight image.  Layer-l fragment layouts (16-byte entries, one per lane):
//   float32 : entry ((kt*MT + mt)*64 + lane) = W_l[o = 16mt + (lane&15)][c = 16kt + 4(lane>>4) + 0..3]
//   fp16-split layer 0 : entry (((ks*MT + mt)*2 + part)*64 + lane) = part (0 high, 1 low) of
//                        W_0[o][c = 32ks + 8(lane>>4) + 0..7] * w_scale[c]
//   (bias column excluded, zero outside the matrix); bias_l[o] = W_l[o][0] when the layer has a bias.
// Layer 0 with a column override (data_transform_obj, BNN_env.py:14-17): an overridden feature column is the
// constant v_c for every row, so its contribution v_c*W0[o][c] moves into the bias and the fragment entry
// becomes 0 - no extra pass over X.
__device__ __forceinline__ void pack_item(int item, const double* __restrict__ w, const double* __restrict__ col_override,
                                          const double* __restrict__ class_w, float* __restrict__ image, const NetMeta& net,
                                          bool with_classw, const float* __restrict__ w_scale = nullptr, int* overflow = nullptr) {
    int piece = item;
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerMeta& L = net.L[l];
        const int n_pieces = L.kt * L.mt * 64;
        if (piece < n_pieces) {
            const int lane = piece & 63;
            const int tile = piece >> 6;
            const int ld = L.in_dim + L.has_bias;
            if (l == 0 && net.l0_f16) {
                const int part = tile & 1, rest = tile >> 1;
                const int mt = rest % L.mt, ks = rest / L.mt;
                const int o = 16 * mt + (lane & 15);
                const int c0 = 32 * ks + 8 * (lane >> 4);
                f16x8 v;
                for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
                if (o < L.out_dim) {
                    const double* row = w + L.w_off + (long long)o * ld + L.has_bias;
                    for (int j = 0; j < 8; ++j) {
                        const int c = c0 + j;
                        if (c < L.in_dim) {
                            const bool overridden = (col_override != nullptr && !isnan(col_override[c]));
                            const float wv = overridden ? 0.f : (float)(row[c] * (double)w_scale[c]);
                            if (overflow && !(fabsf(wv) <= kF16Safe)) *overflow = 1;
                            _Float16 hi, lo;
                            split_f16(wv, hi, lo);
                            v[j] = part ? lo : hi;
                        }
                    }
                }
                *reinterpret_cast<f16x8*>(image + L.frag_off + (long long)piece * 4) = v;
                return;
            }
            const int mt = tile % L.mt, kt = tile / L.mt;
            const int o = 16 * mt + (lane & 15);
            const int c0 = 16 * kt + 4 * (lane >> 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (o < L.out_dim) {
                const double* row = w + L.w_off + (long long)o * ld + L.has_bias;
                for (int s = 0; s < 4; ++s) {
                    const int c = c0 + s;
                    if (c < L.in_dim) {
                        bool overridden = (l == 0 && col_override != nullptr && !isnan(col_override[c]));
                        v[s] = overridden ? 0.f : (float)row[c];
                    }
                }
            }
            *reinterpret_cast<f32x4*>(image + L.frag_off + (long long)piece * 4) = v;
            return;
        }
        piece -= n_pieces;
    }
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerMeta& L = net.L[l];
        const int nb = 16 * L.mt;
        if (piece < nb) {
            const int o = piece;
            double b = 0.0;
            if (o < L.out_dim) {
                const int ld = L.in_dim + L.has_bias;
                const double* row = w + L.w_off + (long long)o * ld;
                if (L.has_bias) b = row[0];
                if (l == 0 && col_override != nullptr) {
                    for (int c = 0; c < L.in_dim; ++c) {
                        const double ov = col_override[c];
                        if (!isnan(ov)) b += ov * row[L.has_bias + c];
                    }
                }
            }
            image[L.bias_off + o] = (float)b;
            return;
        }
        piece -= nb;
    }
    if (with_classw && piece < NPBNN_MAX_WIDTH) {
        image[net.classw_off + piece] = (class_w != nullptr && piece < net.n_out) ? (float)class_w[piece] : 1.0f;
    }
}

__host__ __device__ inline int pack_item_count(const NetMeta& net, bool with_classw) {
    int total = with_classw ? NPBNN_MAX_WIDTH : 0;
    for (int l = 0; l < net.n_layers; ++l) total += net.L[l].kt * net.L[l].mt * 64 + 16 * net.L[l].mt;
    return total;
}

#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) pack_weights_kernel(const double* __restrict__ w, const double* __restrict__ col_override,
                                                           const double* __restrict__ class_w, float* __restrict__ image,
                                                           NetMeta net, const float* __restrict__ w_scale, int* overflow) {
    pack_item(blockIdx.x * 256 + threadIdx.x, w, col_override, class_w, image, net, true, w_scale, overflow);
}
#endif  // NPBNN_KERNELS_MAIN

// ------------------------------------------------------------------------------------------------
// fp16-split copy of the feature matrix (built once per data set, on the device)
//   col_absmax_kernel : per-column max |x| (atomic max on the bit pattern of the non-negative floats)
//   col_scale_kernel  : x_scale[c] = 2^-e, w_scale[c] = 2^e with 2^(e-1) <= max|x_c| < 2^e  (exact powers of two)
//   split_x_kernel    : per row and per 8 features: 8 x fp16 high parts, then 8 x fp16 low parts of x * x_scale
// ------------------------------------------------------------------------------------------------
#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) col_absmax_kernel(const float* __restrict__ X, long long n_rows, int Fp, unsigned* __restrict__ absmax) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Fp) return;
    const long long r0 = (long long)blockIdx.y * 1024;
    long long r1 = r0 + 1024;
    if (r1 > n_rows) r1 = n_rows;
    float m = 0.f;
    for (long long r = r0; r < r1; ++r) {
        const float a = fabsf(X[r * Fp + c]);
        m = (a > m || isnan(a)) ? a : m;
    }
    atomicMax(absmax + c, __float_as_uint(m));     // NaN / inf bit patterns compare above every finite value
}
#endif  // NPBNN_KERNELS_MAIN

#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) col_scale_kernel(const unsigned* __restrict__ absmax, int Fp, float* __restrict__ x_scale,
                                                        float* __restrict__ w_scale) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Fp) return;
    const float m = __uint_as_float(absmax[c]);
    int e = 0;
    if (m > 0.f && isfinite(m)) (void)frexpf(m, &e);
    x_scale[c] = ldexpf(1.f, -e);
    w_scale[c] = ldexpf(1.f, e);
}
#endif  // NPBNN_KERNELS_MAIN

#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) split_x_kernel(const float* __restrict__ X, long long n_rows_pad, int Fp, int Fp16,
                                                      const float* __restrict__ x_scale, float* __restrict__ X16,
                                                      unsigned* __restrict__ absmax_scaled) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;     // one thread per (row, group of 8 features)
    const int groups = Fp16 >> 3;
    if (g >= n_rows_pad * groups) return;
    const long long r = g / groups;
    const int c0 = (int)(g % groups) * 8;
    f16x8 hi, lo;
    float m = 0.f;
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        const float v = c < Fp ? X[r * Fp + c] * x_scale[c] : 0.f;
        _Float16 h, l;
        split_f16(v, h, l);
        hi[j] = h;
        lo[j] = l;
        const float a = fabsf(v);
        m = (a > m || isnan(a)) ? a : m;
    }
    f16x8* dst = reinterpret_cast<f16x8*>(X16 + r * Fp16 + c0);
    dst[0] = hi;
    dst[1] = lo;
    if (m > 1.0f || isnan(m)) atomicMax(absmax_scaled, __float_as_uint(m));   // only a test set scaled by the training scales
}
#endif  // NPBNN_KERNELS_MAIN

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float z, int kind, float prm) {
#ifdef NPBNN_EXP_NO_ACT      // timing experiment only: activation = identity
    return z;
#endif
    switch (kind) {
        case NPBNN_ACT_RELU: return fmaxf(z, 0.f);                                  // BNN_lib.py:51
        case NPBNN_ACT_LEAKY: return z < 0.f ? prm * z : z;                         // BNN_lib.py:55
        case NPBNN_ACT_SWISH: return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));   // BNN_lib.py:60
        default: return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * z) + 1.f);   // BNN_lib.py:65 (exp-form tanh)
    }
}

template <int KIND, int HT>
__device__ __forceinline__ void act_tiles(f32x4 (&h)[HT], int live, float prm) {
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
        if (mt < live)
#pragma unroll
            for (int i = 0; i < 4; ++i) h[mt][i] = act_apply(h[mt][i], KIND, prm);
}

// activation on the first `live` tiles only (wave-uniform kind and count)
template <int HT>
__device__ __forceinline__ void act_live(f32x4 (&h)[HT], int live, int kind, float prm) {
    switch (kind) {
        case NPBNN_ACT_RELU: act_tiles<NPBNN_ACT_RELU>(h, live, prm); break;
        case NPBNN_ACT_LEAKY: act_tiles<NPBNN_ACT_LEAKY>(h, live, prm); break;
        case NPBNN_ACT_SWISH: act_tiles<NPBNN_ACT_SWISH>(h, live, prm); break;
        default: act_tiles<NPBNN_ACT_TANH>(h, live, prm); break;
    }
}

__device__ __forceinline__ float softplus_f(float z) {   // np.logaddexp(0, z), BNN_lib.py:172
    return fmaxf(z, 0.f) + log1pf(__expf(-fabsf(z)));
}

// Reductions over the 4 lanes {l, l^16, l^32, l^48} that hold one data row's units, without LDS traffic:
// v_permlane16_swap exchanges odd and even 16-lane rows, v_permlane32_swap the two 32-lane halves; after a swap of two
// copies of v the pair (r[0], r[1]) holds v[l] and v[l^16] (resp. v[l^32]) in every lane.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float quad_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// arg-max over the quad: larger value wins, ties go to the smaller index (np.argmax takes the first maximum)
__device__ __forceinline__ void quad_argmax(float& bv, int& bi) {
    u32x2 rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    u32x2 ri = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
    {
        const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]);
        const int i0 = (int)ri[0], i1 = (int)ri[1];
        const bool take1 = v1 > v0 || (v1 == v0 && i1 < i0);
        bv = take1 ? v1 : v0;
        bi = take1 ? i1 : i0;
    }
    rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv), __float_as_uint(bv), false, false);
    ri = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
    {
        const float v0 = __uint_as_float(rv[0]), v1 = __uint_as_float(rv[1]);
        const int i0 = (int)ri[0], i1 = (int)ri[1];
        const bool take1 = v1 > v0 || (v1 == v0 && i1 < i0);
        bv = take1 ? v1 : v0;
        bi = take1 ? i1 : i0;
    }
}

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// float64 sums across lanes without LDS traffic (the ds_bpermute behind __shfl_xor costs ~100 cycles per step, two per
// double): rotations inside a 16-lane row by DPP, rows combined with the permlane swaps.  Every lane ends with the sum.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum_f64(double v) {      // over the 16 lanes of a row (row_ror:8, 4, 2, 1)
    v += dpp_f64<0x128>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x122>(v);
    v += dpp_f64<0x121>(v);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {     // over all 64 lanes
    v = row_sum_f64(v);
    u32x2 rl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    u32x2 rh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    rl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    rh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}

#define NPBNN_WAIT_VMCNT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define NPBNN_WAIT_VMCNT(n) NPBNN_WAIT_VMCNT_(n)
#define NPBNN_DEPTH (NPBNN_RING - 1)
#if NPBNN_RING == 4
#define NPBNN_DEPTH_LIT 3
#elif NPBNN_RING == 6
#define NPBNN_DEPTH_LIT 5
#elif NPBNN_RING == 8
#define NPBNN_DEPTH_LIT 7
#else
#error "NPBNN_RING must be 4, 6 or 8"
#endif

__device__ __forceinline__ void wait_younger(int younger) {   // wave-uniform argument; tail / shallow-ring path only
    if (younger >= 7) NPBNN_WAIT_VMCNT(7);
    else if (younger == 6) NPBNN_WAIT_VMCNT(6);
    else if (younger == 5) NPBNN_WAIT_VMCNT(5);
    else if (younger == 4) NPBNN_WAIT_VMCNT(4);
    else if (younger == 3) NPBNN_WAIT_VMCNT(3);
    else if (younger == 2) NPBNN_WAIT_VMCNT(2);
    else if (younger == 1) NPBNN_WAIT_VMCNT(1);
    else NPBNN_WAIT_VMCNT(0);
}
template <int N>
__device__ __forceinline__ void wait_depth() {
    static_assert(N >= 0 && N <= 7, "ring depth");
    if constexpr (N == 7) NPBNN_WAIT_VMCNT(7);
    else if constexpr (N == 6) NPBNN_WAIT_VMCNT(6);
    else if constexpr (N == 5) NPBNN_WAIT_VMCNT(5);
    else if constexpr (N == 4) NPBNN_WAIT_VMCNT(4);
    else if constexpr (N == 3) NPBNN_WAIT_VMCNT(3);
    else if constexpr (N == 2) NPBNN_WAIT_VMCNT(2);
    else if constexpr (N == 1) NPBNN_WAIT_VMCNT(1);
    else NPBNN_WAIT_VMCNT(0);
}
__device__ __forceinline__ int ring_next(int slot) {            // byte offset of the next 1-KiB ring slot
    slot += 1024;
    return slot == kRing * 1024 ? 0 : slot;
}

typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

__device__ __forceinline__ void dma16(const float* g, char* l) {
    __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 4, 0, 0);
}

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
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int n_blocks, int nvals, double* tot /*LDS*/) {
    // partials are laid out [value][workgroup]; wave w of this block sums values w, w+nw, ...: each lane adds workgroups
    // lane, lane+64, ... in order, then a fixed butterfly.  No float atomics -> deterministic.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int v = wave; v < nvals; v += nw) {
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
    const int nvals = (f.lik_kind == NPBNN_LIK_GAUSS) ? kPartialStride : 1;
    reduce_partials(f.partials, f.n_waves, nvals, tot);
    if (threadIdx.x == 0) loglik_from_totals(tot, f.lik_kind, f.k_targets, f.n_rows, f.lik_temp, f.sigma_given, f.sigma, f.out);
}
#endif  // NPBNN_KERNELS_MAIN

// ------------------------------------------------------------------------------------------------
// device-resident Metropolis-Hastings chain
//
// K iterations of MCMC.mh_step (BNN_env.py:381-532, default path: UpdateNormal proposals, BNN_mcmc.py:57-69)
// run as an alternation  step_kernel(t) -> eval_kernel -> step_kernel(t+1) ...  on the chain's stream.  The host
// pre-draws the random numbers of the K iterations (npbnn_host.c) so the proposals are the reference's.
// step_kernel (one workgroup):
//   1. finish iteration t-1: reduce the eval partials -> logLik', accept test
//        (logPost' - logPost) * temperature + hastings >= log u          (BNN_env.py:493-494)
//      on accept commit the changed weights into W_cur, else roll W_prop back;
//   2. propose iteration t: W_prop[idx] = reflect(W_cur[idx] + delta) * mask   (BNN_mcmc.py:64-67, BNN_env.py:461-462);
//   3. logPrior' = sum_l sum log p(W_prop_l; 0, scale_l)                       (npBNN.calc_prior, BNN_env.py:180-194);
//   4. patch the float32 fragment image at the touched entries (w2img map) for the eval kernel.
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
    int pad_;
};

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
    int K, M, D, n_blocks;
    int prior_kind;
    double prior_scale[kMaxLayers];
    double half_inv_s2[kMaxLayers];   // 0.5 / scale^2 (normal prior)
    double w_bound;
    double temperature, lik_temp;
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
    int first;     // first launch of a batch: full prior of the current state, no pass to decide
    int dec;       // parity of the pass to decide, or -1
    int fly;       // parity of the pass being evaluated while this step runs (overlapped schedule), or -1
    int out;       // parity of the pass to prepare
    int launch;    // launch index within the batch (overlapped schedule)
};
__device__ __forceinline__ StepPlan overlapped_plan(int launch) {
    StepPlan pl;
    pl.first = 0;
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
    int s_accepted, s_t, s_start;
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
    if (pl.first && c.prior_kind != NPBNN_PRIOR_UNIFORM) {
        double lp = 0.0;
        for (int l = 0; l < c.net.n_layers; ++l) {
            const LayerMeta& L = c.net.L[l];
            const int n = L.out_dim * (L.in_dim + L.has_bias);
            const double sc = c.prior_scale[l];
            if (c.prior_kind == NPBNN_PRIOR_NORMAL) {
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
    double d_cand[kMaxCand], d_logu[kMaxCand], d_h[kMaxCand], d_ll = 0.0, d_lp = 0.0;
#pragma unroll
    for (int j = 0; j < kMaxCand; ++j) { d_cand[j] = 0.0; d_logu[j] = 0.0; d_h[j] = 0.0; }
    if (tid == 0 && n_pend > 0) {
        d_ll = st->logLik;
        d_lp = st->logPrior;
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
        const int nvals = (lik_kind == NPBNN_LIK_GAUSS) ? kPartialStride : 1;
        {   // wave w sums items w, w+nw, ... (item = candidate * nvals + value): lanes add workgroups lane, lane+64, ... in order
            const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
            const double* part = c.partials + (size_t)pl.dec * part_stride;
            for (int item = wave; item < n_pend * nvals; item += nw) {
                const int j = item / nvals, v = item % nvals;
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
                    loglik_from_totals(sh.tot[j], lik_kind, c.net.k_targets, c.n_rows, c.lik_temp, c.sigma_given, c.sigma_fixed, &sh.o);
                    const double lp = d_cand[j];
                    const double post_new = sh.o.loglik + lp, post_old = d_ll + d_lp;
                    const int a = ((post_new - post_old) * c.temperature + d_h[j] >= d_logu[j]) ? 1 : 0;
                    c.out_acc[t] = (unsigned char)a;
                    c.out_ll[t] = sh.o.loglik;
                    c.out_lp[t] = lp;
                    if (a) {
                        st->logLik = sh.o.loglik;
                        st->logPrior = lp;
                        st->n_accepted += 1;
                        if (lik_kind == NPBNN_LIK_GAUSS)
                            for (int q = 0; q < c.net.k_targets; ++q) st->sigma[q] = sh.o.sigma[q];
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
                    st->void_launch = pl.launch;
                    if (c.pass[pl.fly].n_cand > 0) st->n_void += 1;
                } else {
                    start = c.pass[pl.fly].t0 + c.pass[pl.fly].n_cand;
                }
            }
            sh.s_start = start;
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
    }
    __syncthreads();

    NPBNN_STAMP(4);
    // ---- 2. prepare the next candidates: each is the current state plus its own iteration's perturbation.  Work items
    //      are (candidate, entry) pairs spread over the whole workgroup; the three prior sums share one reduction. ----
    const int t_new = sh.s_start;
    int n_new = c.K - t_new;
    if (n_new > c.D) n_new = c.D;
    if (n_new < 0) n_new = 0;
    double dlp[kMaxCand];
    {
        // staged so that the loads of all candidates are in flight together: (1) the pre-drawn entry, (2) the weight it
        // touches, (3) arithmetic and stores.  One entry per thread and candidate; wider proposals loop.
        const double* __restrict__ wcur = c.w_cur;
        const double* __restrict__ mask = c.mask;
        double* __restrict__ pv_out = c.pv + (size_t)pl.out * pv_stride;
        int woff[kMaxLayers];
        double half_inv_s2[kMaxLayers];
#pragma unroll
        for (int q = 0; q < kMaxLayers; ++q) {
            woff[q] = q < c.net.n_layers ? c.net.L[q].w_off : 0x7fffffff;
            half_inv_s2[q] = c.half_inv_s2[q];
        }
        int ii[kMaxCand], pp[kMaxCand];
        double dd[kMaxCand], bb[kMaxCand], mm[kMaxCand];
        float ss[kMaxCand];
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
            dlp[j] = 0.0;
            ii[j] = -1; pp[j] = 0; dd[j] = 0.0; ss[j] = 1.0f;
            if (j < n_new && tid < c.cnt[t_new + j]) {
                const size_t k = (size_t)(t_new + j) * c.M + tid;
                ii[j] = c.idx[k];
                dd[j] = c.delta[k];
                pp[j] = c.pos[k];
                if (c.pscale) ss[j] = c.pscale[k];
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
            bb[j] = ii[j] >= 0 ? wcur[ii[j]] : 0.0;
            mm[j] = (ii[j] >= 0 && mask) ? mask[ii[j]] : 1.0;
        }
        auto make = [&](int j, int e, int i, double base, double d, double m, int pos, float sc) {
            double v = base + d;
            if (v > c.w_bound) v = c.w_bound - (v - c.w_bound);
            if (v < -c.w_bound) v = -c.w_bound + (-c.w_bound - v);
            v *= m;
            pv_out[(size_t)j * c.M + e] = v;
            if (pos < 0 && !(fabs(v * (double)sc) <= (double)kF16Safe)) *c.overflow = 1;
            if (c.prior_kind != NPBNN_PRIOR_UNIFORM) {
                int l = 0;
#pragma unroll
                for (int q = 1; q < kMaxLayers; ++q) l += (i >= woff[q]) ? 1 : 0;
                if (c.prior_kind == NPBNN_PRIOR_NORMAL) dlp[j] -= (v * v - base * base) * half_inv_s2[l];
                else dlp[j] += prior_delta(c.prior_kind, v, base, c.prior_scale[l]);
            }
        };
#pragma unroll
        for (int j = 0; j < kMaxCand; ++j) {
            if (ii[j] >= 0) make(j, tid, ii[j], bb[j], dd[j], mm[j], pp[j], ss[j]);
            if (j < n_new) {
                const size_t row = (size_t)(t_new + j) * c.M;
                for (int e = tid + blockDim.x; e < c.cnt[t_new + j]; e += blockDim.x) {
                    const int i = c.idx[row + e];
                    if (i >= 0) make(j, e, i, wcur[i], c.delta[row + e], mask ? mask[i] : 1.0, c.pos[row + e], c.pscale ? c.pscale[row + e] : 1.0f);
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
        PassDesc d;
        d.t0 = t_new;
        d.n_cand = n_new;
        for (int j = 0; j < kMaxCand; ++j) {
            double sj = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sj += sh.red3[j][w];
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

#ifdef NPBNN_KERNELS_MAIN
// serial schedule: the step as a kernel of its own, between two evaluation kernels (and as the first launch of every batch)
__global__ void __launch_bounds__(1024) chain_step_kernel(const ChainParams* __restrict__ cp, int first_launch) {
    const ChainParams& c = *cp;           // device-resident parameter block; only the per-launch scalar travels as an argument
    __shared__ StepShared sh;
    if (!first_launch && c.pass[0].n_cand == 0) return;      // launched past the end of the batch
    StepPlan pl;
    pl.first = first_launch;
    pl.dec = first_launch ? -1 : 0;
    pl.fly = -1;
    pl.out = 0;
    pl.launch = 0;
    chain_step(c, pl, sh);
}
#endif  // NPBNN_KERNELS_MAIN

// ------------------------------------------------------------------------------------------------
// fused forward + likelihood
//   MT0 : 16-unit tiles of layer 0's output (accumulators of the streamed GEMM)
//   MTI : max 16-unit tiles of any later layer's output (1 covers every net whose hidden layers after the first
//         and whose output have <= 16 nodes - all BASELINE configs; 8 is the general case)
//   F16 : fp16-split layer 0
//   D   : weight sets ("candidates") evaluated against one streaming read of X (speculative chain passes; 1 otherwise)
//   LK  : likelihood class the epilogue is built for - kLikCat (categorical / none), kLikGauss (residual moments) or
//         kLikGen (float64 row-wise likelihoods: predicted sigma, Poisson, negative binomial).  Separate builds because
//         each class keeps different per-lane accumulators alive through the whole kernel (and lgamma is register hungry).
// ------------------------------------------------------------------------------------------------
typedef void (*eval_fn_t)(const EvalParams*, int);
constexpr int kLikCat = 0, kLikGauss = 1, kLikGen = 2;
__host__ __device__ inline int lik_class(int lik_kind) {
    return lik_needs_row_scratch(lik_kind) ? kLikGen : (lik_kind == NPBNN_LIK_GAUSS ? kLikGauss : kLikCat);
}

template <int LK>
struct TileAcc {            // per-candidate float64 accumulators of one wave: sum of the per-row log-likelihood terms ...
    double ll;
};
template <>
struct TileAcc<kLikGauss> { // ... or, for the Gaussian likelihood, residual moments of the 4 target columns this lane owns
    double s1[4], s2[4];
};

// A value every lane of the wave holds identically, moved to scalar registers.  The kernel reads its launch-invariant
// parameters through pointers that other code inlined into it (the chain step) writes through, and its first branch is
// lane dependent (the diagnostic stamps), so the compiler no longer proves them uniform by itself - and a loop bound it
// believes divergent turns every branch of the main loop into exec-mask bookkeeping.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ long long uni(long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ T* uni(T* ptr) { return reinterpret_cast<T*>(uni((long long)reinterpret_cast<unsigned long long>(ptr))); }

// Launch-invariant scalars of the parameter block, copied once so that they stay in SGPRs: the counted s_waitcnt
// statements of the main loop are memory clobbers, and anything read through the block pointer would be fetched again
// after each of them.
struct HotParams {
    const int* labels;
    const float* targets;
    const float* inst_w;
    unsigned* confusion;
    float* y_out;
    long long n_rows;
    int use_classw, predict_mode, weight_sets;
    int n_layers, C, MTL, lik_kind, k_targets, act_kind, out_kind, final_act, classw_off;
};

template <int KIND, int HT, int D>
__device__ __forceinline__ void act_tiles_all(f32x4 (&h)[D][HT], int live, float prm) {
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
        if (mt < live)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) h[j][mt][i] = act_apply(h[j][mt][i], KIND, prm);
}
// activation on the first `live` tiles of every candidate (wave-uniform kind and count; candidates innermost so that
// their independent exp / rcp chains interleave)
template <int HT, int D>
__device__ __forceinline__ void act_live_all(f32x4 (&h)[D][HT], int live, int kind, float prm) {
    switch (kind) {
        case NPBNN_ACT_RELU: act_tiles_all<NPBNN_ACT_RELU>(h, live, prm); break;
        case NPBNN_ACT_LEAKY: act_tiles_all<NPBNN_ACT_LEAKY>(h, live, prm); break;
        case NPBNN_ACT_SWISH: act_tiles_all<NPBNN_ACT_SWISH>(h, live, prm); break;
        default: act_tiles_all<NPBNN_ACT_TANH>(h, live, prm); break;
    }
}

// layers 1..L-1 and the likelihood epilogue of one 16-row tile for the D candidates of the pass (weight images
// `imgs + j*image_floats` in LDS).  Every stage loops over the candidates innermost: their chains (dependent MFMAs,
// exp / rcp / log) are independent, so the wave always has three of them to interleave.
// Handles candidates J0 .. J0+D-1 of the DA the pass holds (all of them when the registers allow, else one at a time).
template <int MT0, int MTI, int LK, int D, int DA, int J0>
__device__ __forceinline__ void tile_tail(const NetMeta& net, const HotParams& hp, const float* imgs0, int image_floats,
                                          const f32x4 (&acc0_all)[DA][MT0], int lane, int n, int kq, const char* a_slot, float* row_scratch,
                                          long long row, bool row_ok, TileAcc<LK> (&A_all)[DA]) {
    static_assert(J0 + D <= DA, "candidate range");
    constexpr bool primary = J0 == 0;              // statistics and predictions come from the first candidate
    const float* const imgs = imgs0 + (size_t)J0 * image_floats;
    auto A = [&](int j) -> TileAcc<LK>& { return A_all[J0 + j]; };
    constexpr int HT = MT0 > MTI ? MT0 : MTI;      // tiles of the widest activation vector held in registers
    const int n_layers = hp.n_layers;
    const int C = hp.C;
    const int MTL = hp.MTL;
    const int lik_kind = hp.lik_kind;
    const int k_targets = hp.k_targets;
    const bool need_softmax = LK == kLikCat && ((lik_kind == NPBNN_LIK_CATEGORICAL) || (hp.predict_mode == 2 && hp.out_kind == NPBNN_OUT_SOFTMAX));
    // ---------------- layers 1..L-1 chained through the accumulators ----------------
    f32x4 h[D][HT];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) h[j][mt] = mt < MT0 ? acc0_all[J0 + j][mt < MT0 ? mt : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 1; l < n_layers; ++l) {
        const LayerMeta& L = net.L[l];
        const int lkt = uni(L.kt), lmt = uni(L.mt);
        act_live_all(h, lkt, hp.act_kind, uni(net.act_prm[l - 1]));
        const float* frag = imgs + uni(L.frag_off) + lane * 4;
        const float* bias = imgs + uni(L.bias_off) + 4 * kq;
        f32x4 acc[D][MTI];
#pragma unroll
        for (int mt = 0; mt < MTI; ++mt) {
#pragma unroll
            for (int j = 0; j < D; ++j) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mt < lmt) {
#pragma unroll
                for (int j = 0; j < D; ++j) acc[j][mt] = *reinterpret_cast<const f32x4*>(bias + (size_t)j * image_floats + 16 * mt);
#pragma unroll
                for (int ct = 0; ct < HT; ++ct) {
                    if (ct < lkt) {
                        f32x4 a[D];
#pragma unroll
                        for (int j = 0; j < D; ++j)
                            a[j] = *reinterpret_cast<const f32x4*>(frag + (size_t)j * image_floats + (size_t)(ct * lmt + mt) * 256);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int j = 0; j < D; ++j)
                                acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][s], h[j][ct][s], acc[j][mt], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int mt = 0; mt < MTI; ++mt) h[j][mt] = acc[j][mt];
    }
    if (hp.final_act) act_live_all(h, MTL, hp.act_kind, uni(net.act_prm[n_layers - 1]));
    // h[j][mt][i] = last-layer value of unit o = 16mt + 4kq + i for data row tile*16 + n   (mt < MTL <= MTI)

    // ---------------- epilogue ----------------
    float lse[D];
    int best_i = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) lse[j] = 0.f;
    if (need_softmax) {
        float m[D], se[D];
#pragma unroll
        for (int j = 0; j < D; ++j) { m[j] = -INFINITY; se[j] = 0.f; }
#pragma unroll
        for (int mt = 0; mt < MTI; ++mt)
            if (mt < MTL)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (16 * mt + 4 * kq + i < C)
#pragma unroll
                        for (int j = 0; j < D; ++j) m[j] = fmaxf(m[j], h[j][mt][i]);
#pragma unroll
        for (int j = 0; j < D; ++j) m[j] = quad_max(m[j]);
#pragma unroll
        for (int mt = 0; mt < MTI; ++mt)
            if (mt < MTL)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (16 * mt + 4 * kq + i < C)
#pragma unroll
                        for (int j = 0; j < D; ++j) se[j] += __expf(h[j][mt][i] - m[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) se[j] = quad_sum(se[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) lse[j] = m[j] + __logf(se[j]);
        if (hp.confusion && primary) {   // np.argmax: first maximum wins (BNN_lib.py:207); statistics of the first candidate only
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int mt = 0; mt < MTI; ++mt)
                if (mt < MTL)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int o = 16 * mt + 4 * kq + i;
                        if (o < C && h[0][mt][i] > bv) { bv = h[0][mt][i]; bi = o; }
                    }
            quad_argmax(bv, bi);
            best_i = bi;
        }
    }

    if constexpr (LK == kLikCat) {
      if (lik_kind == NPBNN_LIK_CATEGORICAL) {
        const int lab = *reinterpret_cast<const int*>(a_slot + n * 4);
        float zl[D];
        bool own = false;
#pragma unroll
        for (int j = 0; j < D; ++j) zl[j] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MTI; ++mt)
            if (mt < MTL)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (16 * mt + 4 * kq + i == lab) {
                        own = true;
#pragma unroll
                        for (int j = 0; j < D; ++j) zl[j] = h[j][mt][i];
                    }
        if (lab >= 0) {
            float wgt = 1.f;
            if (hp.inst_w) wgt *= *reinterpret_cast<const float*>(a_slot + 64 + n * 4);
            if (hp.use_classw) wgt *= imgs[hp.classw_off + lab];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float term = 0.f;
                if (own) term += zl[j];
                if (kq == 0) term -= lse[j];
                term *= wgt;
                A(j).ll += (double)term;
            }
            if (hp.confusion && primary && kq == 0 && best_i < C) atomicAdd(hp.confusion + lab * C + best_i, 1u);
        }
      }
    } else if constexpr (LK == kLikGen) {
        // (kLikGen builds only: the float64 lgamma / log / exp below would otherwise cost the hot kernels their registers)
        // likelihoods pairing output j with output k+j of the same row (BNN_lib.py:134-143, BNN_lik.py:5-66): the 16
        // outputs of a row meet through LDS; lane (n, kq) then owns target columns j = kq, kq+4, ...; float64 terms
        const float* tg = reinterpret_cast<const float*>(a_slot + 128);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            *reinterpret_cast<f32x4*>(row_scratch + n * 16 + 4 * kq) = h[c][0];
            double term = 0.0;
            if (row_ok) {
                for (int j = kq; j < k_targets; j += 4) {
                    const double y = (double)tg[n * k_targets + j];
                    if (lik_kind == NPBNN_LIK_GAUSS_PRED_SIGMA) {
                        const double mu = (double)row_scratch[n * 16 + j];
                        const double zs = (double)row_scratch[n * 16 + k_targets + j];
                        const double sg = fmax(zs, 0.0) + log1p(exp(-fabs(zs)));      // softplus, BNN_lib.py:172,181
                        const double r = (y - mu) / sg;
                        term += -0.9189385332046727418 - log(sg) - 0.5 * r * r;
                    } else if (lik_kind == NPBNN_LIK_POISSON) {
                        if (j == 0) {
                            const double eta = (double)row_scratch[n * 16];
                            term += y * eta - exp(eta) - lgamma(y + 1.0);             // poisson.logpmf(k, exp(eta))
                        }
                    } else {
                        const bool one_col = lik_kind != NPBNN_LIK_NEGBIN2D;
                        if (one_col && j > 0) continue;
                        const int jp = one_col ? 1 : k_targets + j;
                        const double e0 = (double)row_scratch[n * 16 + j], e1 = (double)row_scratch[n * 16 + jp];
                        double mean, pr;
                        if (lik_kind == NPBNN_LIK_NEGBIN_BASE10) {
                            mean = exp(2.302585092994046 * e0);
                            pr = 1.0 / (1.0 + exp(-2.302585092994046 * e1));
                        } else {
                            mean = exp(e0);
                            pr = 1.0 / (1.0 + exp(-e1));
                        }
                        const double nn = pr * mean / (1.0 - pr);
                        // nbinom.logpmf(k; n, p) = lgamma(k+n) - lgamma(k+1) - lgamma(n) + n log p + k log(1-p)
                        term += lgamma(y + nn) - lgamma(y + 1.0) - lgamma(nn) + nn * log(pr) + y * log1p(-pr);
                    }
                }
            }
            A(c).ll += term;
        }
    } else {
        const float* tg = reinterpret_cast<const float*>(a_slot + 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 4 * kq + i;
            if (o < k_targets && row_ok) {
                const float y = tg[n * k_targets + o];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float r = y - h[j][0][i];
                    A(j).s1[i] += (double)r;
                    A(j).s2[i] += (double)r * (double)r;
                }
            }
        }
    }

    if (hp.predict_mode && row_ok) {       // predictions: of the first candidate, or of every weight set of the launch
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (!(hp.weight_sets || (primary && j == 0))) continue;
            float* const yo = hp.y_out + (size_t)(J0 + j) * (size_t)hp.n_rows * C;
#pragma unroll
            for (int mt = 0; mt < MTI; ++mt)
                if (mt < MTL)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int o = 16 * mt + 4 * kq + i;
                        if (o < C) {
                            float v = h[j][mt][i];
                            if (hp.predict_mode == 2) {
                                if (hp.out_kind == NPBNN_OUT_SOFTMAX) v = __expf(v - lse[j]);
                                else if (hp.out_kind == NPBNN_OUT_SOFTPLUS_HALF && o >= C / 2) v = softplus_f(v);
                            }
                            yo[row * C + o] = v;
                        }
                    }
        }
    }
}

// waves per workgroup a build is compiled for: more candidates keep more accumulators and weight fragments alive
__host__ __device__ constexpr int max_waves_for(int mti, int d) { return mti == 1 ? (d == 1 ? 16 : d == 2 ? 14 : 11) : 8; }
// software-pipelined layer 0 (the fragments of K-step s+1 are read from LDS while the MFMAs of step s run): builds whose two
// fragment sets fit the register budget of their launch bounds
__host__ __device__ constexpr bool pipelined_l0(int mt0, int mti, bool f16, int d) {
    return f16 && mti == 1 && ((d == 3 && mt0 <= 2) || (d == 2 && mt0 <= 1) || (d == 1 && mt0 <= 3));
}

#define NPBNN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int MT0, int MTI, bool F16, int D, int LK>
__global__ void __launch_bounds__(max_waves_for(MTI, D) * 64) eval_kernel(const EvalParams* __restrict__ pp, int launch) {
    // the parameter block lives in device memory (warm in L2 across the thousands of launches of a chain); a by-value
    // kernel argument of this size costs several microseconds of cold scalar loads per launch
    const EvalParams& p = *pp;
    const int bid = uni((int)blockIdx.x);      // (pinned to a scalar register before the first lane-dependent branch)
    unsigned long long* const stamps = uni(p.stamps);
#define NPBNN_ESTAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)bid * 8 + (k)] = wall_clock64(); } while (0)
    NPBNN_ESTAMP(0);
    constexpr int DEPTH = F16 ? ((kRing - 1) & ~1) : kRing - 1;   // pieces in flight; whole pairs in fp16-split mode
    constexpr bool PIPE = pipelined_l0(MT0, MTI, F16, D) && kRing == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // overlapped chain schedule: the last workgroup decides the previous pass and prepares the next one while the others
    // evaluate this one (chain_step above); passes alternate between two sets of descriptors / patch values / partial sums
    const ChainParams* const chain = uni(p.chain);
    const int G = (int)gridDim.x - (chain ? 1 : 0);     // workgroups that evaluate
    if (chain && bid == G) {
        chain_step(*chain, overlapped_plan(launch), *reinterpret_cast<StepShared*>(smem));
        return;
    }
    const int par = chain ? (launch & 1) : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const NetMeta& net = p.net;
    const int wpb = blockDim.x >> 6;
    const int image_floats = uni(net.image_floats);
    const size_t IB = (size_t)image_floats * 4;                     // bytes of one weight image

    HotParams hp;
    hp.labels = uni(p.labels); hp.targets = uni(p.targets); hp.inst_w = uni(p.inst_w); hp.confusion = uni(p.confusion);
    hp.y_out = uni(p.y_out);
    hp.n_rows = uni(p.n_rows); hp.use_classw = uni(p.use_classw); hp.predict_mode = uni(p.predict_mode);
    hp.weight_sets = uni(p.weight_sets);
    hp.n_layers = uni(net.n_layers); hp.C = uni(net.n_out); hp.MTL = uni(net.L[hp.n_layers - 1].mt); hp.lik_kind = uni(net.lik_kind);
    hp.k_targets = uni(net.k_targets); hp.act_kind = uni(net.act_kind); hp.out_kind = uni(net.out_kind);
    hp.final_act = uni(net.final_act);
    hp.classw_off = uni(net.classw_off);
    const int k_targets = hp.k_targets;
    const int aux_sz = aux_bytes(k_targets);
    const float* const Xg = uni(p.X);
    const int Fp = uni(p.Fp);
    const int n_tiles = uni(p.n_tiles);
    const int M = uni(p.M);
    const int* const g_pos = uni(p.pos);
    const float* const g_pscale = uni(p.pscale);
    double* const g_partials = uni(p.partials);

    // ---- which candidates does this pass evaluate?  A chain pass always computes all D weight sets (the step kernel
    //      pads the tail of a batch with unperturbed copies, cnt = 0, whose sums nobody reads): no per-candidate branches ----
    int t0 = 0;
    int cnt[D];
#pragma unroll
    for (int j = 0; j < D; ++j) cnt[j] = 0;
    const PassDesc* const pass = uni(p.has_pass) ? &p.pass_desc[par] : nullptr;
    const double* const pv = uni(p.pv) + (size_t)par * kMaxCand * M;
    if (pass) {
        if (uni(pass->n_cand) == 0) return;                         // the chain batch is finished
        t0 = uni(pass->t0);
#pragma unroll
        for (int j = 0; j < D; ++j) cnt[j] = uni(pass->cnt[j < kMaxCand ? j : 0]);
    }

    char* const ring = smem + D * IB + (size_t)wave * wave_lds_bytes(k_targets, hp.lik_kind);
    char* const aux = ring + kRing * 1024;
    float* const row_scratch = reinterpret_cast<float*>(aux + kAuxSlots * aux_sz);   // [16 rows][16 outputs], generic likelihoods

    // ---- stage the weight image of the current state into LDS, once per candidate: lane-linear DMA copies ----
    {
        const int n_pieces = image_floats >> 8;   // 1-KiB pieces
        const float* const image = uni(p.image);
        const size_t set_stride = hp.weight_sets ? (size_t)image_floats : 0;
#pragma unroll
        for (int j = 0; j < D; ++j)
            for (int i = wave; i < n_pieces; i += wpb)
                dma16(image + j * set_stride + (size_t)i * 256 + lane * 4, smem + j * IB + (size_t)i * 1024);
    }

    // ---- tile schedule: workgroup b owns tiles b, b+G, b+2G, ...; its m-th tile goes to wave m % wpb, so the
    //      tile counts of the waves (and SIMDs) of one CU differ by at most one ----
    const int KT0 = uni(net.L[0].kt);
    const int first_tile = bid + G * wave;
    const int stride = G * wpb;
    const int my_tiles = first_tile < n_tiles ? (n_tiles - first_tile + stride - 1) / stride : 0;
    const int Q = my_tiles * KT0;                       // 1-KiB X pieces this wave consumes
    int Dp = PIPE ? kRing : DEPTH;                      // prefetch distance in pieces
    if (Dp > 2 * KT0) Dp = 2 * KT0;                     // at most 3 tiles in flight (aux slots)
    const bool full_depth = (Dp == DEPTH);

    // prefetch cursor: a per-lane running source pointer and a scalar ring offset
    const float* pf_ptr = Xg + ((size_t)first_tile * 16 + n) * (size_t)Fp + 4 * kq;
    const size_t tile_jump = (size_t)stride * 16 * (size_t)Fp - (size_t)KT0 * 16;
    int pf_q = 0, pf_kt = 0, pf_tile = first_tile, pf_seq = 0, pf_slot = 0;
    auto issue_aux = [&]() {   // row-aux data of a tile travels ahead of its first X piece
        char* a = aux + (pf_seq & (kAuxSlots - 1)) * aux_sz;
        const size_t r0 = (size_t)pf_tile * 16;
        if (lane < 16) {
            if (hp.labels) dma4(hp.labels + r0 + lane, a);
            if (hp.inst_w) dma4(hp.inst_w + r0 + lane, a + 64);
        }
        if (hp.targets) {
            const int total = 16 * k_targets;           // contiguous floats of this tile's targets
            for (int e = 0; e < total; e += 64) {
                const int idx = e + lane;               // (only the lanes with an element take part: an LDS-DMA writes
                if (idx < total)                        //  lane*4 bytes past its base whatever it loaded, and the slot ends at `total`)
                    dma4(hp.targets + r0 * k_targets + idx, a + 128 + e * 4);
            }
        }
    };
    auto issue_next = [&]() {
        if (pf_kt == 0) issue_aux();
        dma16(pf_ptr, ring + pf_slot);
        pf_ptr += 16;
        pf_slot = ring_next(pf_slot);
        ++pf_q;
        if (++pf_kt == KT0) { pf_kt = 0; pf_tile += stride; ++pf_seq; pf_ptr += tile_jump; }
    };
    for (int i = 0; i < Dp && pf_q < Q; ++i) issue_next();

    // ---- candidates = current state + their own touched entries: fetch the first entry per thread now (its latency
    //      hides under the image copy), meet, patch the LDS images, meet again.  The first barrier also waits for this
    //      wave's image pieces and first X pieces (needed next anyway). ----
    int ppos[D];
    double pval[D];
    float psc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        ppos[j] = 0; pval[j] = 0.0; psc[j] = 1.0f;
        if (pass && tid < cnt[j]) {
            const size_t k = (size_t)(t0 + j) * M + tid;
            ppos[j] = g_pos[k];
            pval[j] = pv[(size_t)j * M + tid];
            if (g_pscale) psc[j] = g_pscale[k];
        }
    }
    NPBNN_ESTAMP(1);
    __syncthreads();
    NPBNN_ESTAMP(2);
    if (pass) {
        auto patch = [&](int j, int pos, double v, float sc) {
            if (pos == 0x7fffffff) return;                  // superseded entry (a later draw of the same position wins)
            float* imgj = reinterpret_cast<float*>(smem + j * IB);
            if (pos < 0) {                               // fp16-split layer-0 entry
                _Float16 hi, lo;
                split_f16((float)(v * (double)sc), hi, lo);
                _Float16* i16 = reinterpret_cast<_Float16*>(imgj);
                const int hpos = pos & 0x7fffffff;
                i16[hpos] = hi;
                i16[hpos + 512] = lo;
            } else {
                imgj[pos] = (float)v;
            }
        };
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (tid < cnt[j]) patch(j, ppos[j], pval[j], psc[j]);
            for (int e = tid + blockDim.x; e < cnt[j]; e += blockDim.x) {
                const size_t k = (size_t)(t0 + j) * M + e;
                patch(j, g_pos[k], pv[(size_t)j * M + e], g_pscale ? g_pscale[k] : 1.0f);
            }
        }
        __syncthreads();
    }
    NPBNN_ESTAMP(3);

    TileAcc<LK> A[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if constexpr (LK == kLikGauss) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { A[j].s1[i] = 0.0; A[j].s2[i] = 0.0; }
        } else {
            A[j].ll = 0.0;
        }
    }

    const float* const imgs = reinterpret_cast<const float*>(smem);
    const int frag0_off = uni(net.L[0].frag_off) + lane * 4;       // float offsets inside an image
    const int bias0_off = uni(net.L[0].bias_off) + 4 * kq;
    auto load_bias0 = [&](f32x4 (&acc0)[D][MT0]) {
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int mt = 0; mt < MT0; ++mt)
                acc0[j][mt] = *reinterpret_cast<const f32x4*>(imgs + (size_t)j * image_floats + bias0_off + 16 * mt);
    };
    auto run_tail = [&](const f32x4 (&acc0)[D][MT0], int tseq, int tile) {
#ifdef NPBNN_EXP_NO_TAIL      // timing experiment only: keep the layer-0 result alive, skip layers 1.. and the likelihood
        if constexpr (LK != kLikGauss) {
#pragma unroll
            for (int j = 0; j < D; ++j) A[j].ll += (double)acc0[j][0][0];
        }
        return;
#endif
        const char* a_slot = aux + (tseq & (kAuxSlots - 1)) * aux_sz;
        const long long row = (long long)tile * 16 + n;
        // the candidates go through the tail together (their independent chains interleave) while the registers allow
        constexpr int HT = MT0 > MTI ? MT0 : MTI;
        constexpr int DT = (D * HT <= 6 && LK != kLikGen) ? D : 1;
        if constexpr (DT == D) {
            tile_tail<MT0, MTI, LK, D, D, 0>(net, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
        } else {
            tile_tail<MT0, MTI, LK, 1, D, 0>(net, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
            if constexpr (D > 1) tile_tail<MT0, MTI, LK, 1, D, 1>(net, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
            if constexpr (D > 2) tile_tail<MT0, MTI, LK, 1, D, 2>(net, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
            static_assert(D <= 3, "add a call per candidate");
        }
    };

    if constexpr (PIPE) {
        // ---------------- fp16-split layer 0, software pipelined over the K-steps of ALL tiles of this wave ----------------
        // one K=32 step = two 1-KiB pieces; lane (n, kg) takes feature group kg: piece kg>>1, entries 2(kg&1) (high parts)
        // and 2(kg&1)+1 (low parts); three MFMAs per unit tile and candidate: wh.xh + wl.xh + wh.xl.
        // Ring = 2 steps.  Per step s: the fragments of step s are in registers (so its two slots are free) -> DMA of step
        // s+2 into them -> wait for step s+1 -> read its fragments -> MFMAs of step s (the LDS reads complete underneath).
        // Work unit = (K-step s, candidate j): 3*MT0 MFMAs on the x fragments of the step and the weight fragments of the
        // candidate.  While unit u computes, the fragments of unit u+1 are read from LDS into the other register set.
        struct WFrag { f16x8 wh[MT0], wl[MT0]; };
        struct XFrag { f16x8 xh, xl; };
        const int KS = KT0 >> 1;                          // K-steps per tile
        const int S = my_tiles * KS;
        if (S > 0) {
            WFrag Wb[2];
            XFrag Xb[2];
            f32x4 acc0[D][MT0];
            int ld_slot = 0;                              // ring offset of the next step to read
            int s = 0, ks = 0;                            // current step: global index, index inside its tile
            auto load_x = [&](XFrag& x) {
                const int slot_b = ring_next(ld_slot);
                const char* px = ring + ((kq >> 1) ? slot_b : ld_slot) + ((2 * (kq & 1)) * 16 + n) * 16;
                x.xh = *reinterpret_cast<const f16x8*>(px);
                x.xl = *reinterpret_cast<const f16x8*>(px + 256);
                ld_slot = ring_next(slot_b);
            };
            auto load_w = [&](WFrag& w, int kstep, int j) {
                const float* fr = imgs + (size_t)j * image_floats + frag0_off + kstep * (MT0 * 512);
#pragma unroll
                for (int mt = 0; mt < MT0; ++mt) {
                    w.wh[mt] = *reinterpret_cast<const f16x8*>(fr + mt * 512);
                    w.wl[mt] = *reinterpret_cast<const f16x8*>(fr + mt * 512 + 256);
                }
            };
            NPBNN_WAIT_VMCNT(0);                          // (the barriers above already drained this wave's loads)
            load_x(Xb[0]);
            load_w(Wb[0], 0, 0);
            load_bias0(acc0);
            // one K-step; PAR = which x set holds it.  Unit j reads its weights from Wb[(PAR*D + j) & 1].
            auto step = [&](auto par_tag) {
                constexpr int PAR = decltype(par_tag)::value;
                const int ks_next = (ks + 1 == KS) ? 0 : ks + 1;
                bool issued = false;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const WFrag& wc = Wb[(PAR * D + j) & 1];
                    WFrag& wn = Wb[(PAR * D + j + 1) & 1];
                    NPBNN_WAIT_LGKM0();                   // this unit's fragments are complete
                    if (j == 0 && pf_q < Q) {             // the x fragments of step s are in registers: refill its slots (step s+2)
                        issue_next();
                        issue_next();
                        issued = true;
                    }
                    if (j == D - 1) {
                        if (s + 1 < S) {
                            if (issued) wait_depth<2>();  // step s+1 has landed
                            else NPBNN_WAIT_VMCNT(0);
                            load_x(Xb[PAR ^ 1]);
                            load_w(wn, ks_next, 0);
                        }
                    } else {
                        load_w(wn, ks, j + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);    // keep the LDS reads of the next unit ahead of this unit's MFMAs
#ifndef NPBNN_EXP_NO_L0       // (timing experiment only: without the layer-0 MFMAs)
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.wh[mt], Xb[PAR].xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.wl[mt], Xb[PAR].xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.wh[mt], Xb[PAR].xl, acc0[j][mt], 0, 0, 0);
#else
                    acc0[j][0][0] += (float)wc.wh[0][0] + (float)wc.wl[MT0 - 1][7] + (float)Xb[PAR].xh[0] + (float)Xb[PAR].xl[7];
#endif
                }
                ++s;
                ks = ks_next;
            };
            int tile = first_tile;
            unsigned long long tail_ticks = 0;
            for (int tseq = 0; tseq < my_tiles; ++tseq, tile += stride) {
                for (int kp = 0; kp + 1 < KS; kp += 2) { // the register sets alternate, no copies
                    step(std::integral_constant<int, 0>{});
                    step(std::integral_constant<int, 1>{});
                }
                if (KS & 1) {                             // odd number of steps per tile: put the next tile's first fragments
                    step(std::integral_constant<int, 0>{});   // back into set 0 (once per tile)
                    Xb[0] = Xb[1];
                    if (D & 1) Wb[0] = Wb[1];
                }
                const unsigned long long tk = stamps ? wall_clock64() : 0;
                run_tail(acc0, tseq, tile);              // (the first fragments of the next tile arrive underneath)
                load_bias0(acc0);
                if (stamps) tail_ticks += wall_clock64() - tk;
            }
            if (stamps && tid == 0) stamps[(size_t)bid * 8 + 7] = tail_ticks;      // diagnostics: time this wave spent in tails
        }
    } else {
    int q = 0, cs_slot = 0;
    int tile = first_tile;
    for (int tseq = 0; tseq < my_tiles; ++tseq, tile += stride) {
        // ---------------- layer 0: H0^T = W0 . X^T, K streamed from the ring, every candidate on the same X piece ----------------
        f32x4 acc0[D][MT0];
        load_bias0(acc0);
        int fr_off = frag0_off;
        auto consume = [&]() {
            if constexpr (F16) {
                // one K=32 step = two 1-KiB pieces; lane (n, kg) takes feature group kg: piece kg>>1, entries 2(kg&1) (high
                // parts) and 2(kg&1)+1 (low parts); three MFMAs per tile: wh.xh + wl.xh + wh.xl
                const int slot_b = ring_next(cs_slot);
                const char* px = ring + ((kq >> 1) ? slot_b : cs_slot) + ((2 * (kq & 1)) * 16 + n) * 16;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(px);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(px + 256);
                cs_slot = ring_next(slot_b);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float* fr = imgs + (size_t)j * image_floats + fr_off;
                    f16x8 wh[MT0], wl[MT0];
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) {
                        wh[mt] = *reinterpret_cast<const f16x8*>(fr + mt * 512);
                        wl[mt] = *reinterpret_cast<const f16x8*>(fr + mt * 512 + 256);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt], xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mt], xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt], xl, acc0[j][mt], 0, 0, 0);
                }
                fr_off += MT0 * 512;
            } else {
                const f32x4 x = *reinterpret_cast<const f32x4*>(ring + cs_slot + lane * 16);
                cs_slot = ring_next(cs_slot);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float* fr = imgs + (size_t)j * image_floats + fr_off;
                    f32x4 a[MT0];
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(fr + mt * 256);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int mt = 0; mt < MT0; ++mt)
                            acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s], x[s], acc0[j][mt], 0, 0, 0);
                }
                fr_off += MT0 * 256;
            }
        };
        constexpr int STEP = F16 ? 2 : 1;               // pieces per consume()
        if (full_depth) {
            // steady part: every step consumed is replaced by one issued -> exactly DEPTH younger pieces in flight;
            // once the wave's last piece has been issued, drain once and consume what is left without waiting
            int n_issue = Q - pf_q;
            if (n_issue > KT0) n_issue = KT0;
            for (int kt = 0; kt < n_issue; kt += STEP) {
                issue_next();                           // targets the slot(s) consumed one step ago
                if constexpr (F16) issue_next();
                wait_depth<DEPTH>();
                consume();
            }
            if (n_issue < KT0) {
                NPBNN_WAIT_VMCNT(0);
                for (int kt = n_issue; kt < KT0; kt += STEP) consume();
            }
            q += KT0;
        } else {
            for (int kt = 0; kt < KT0; kt += STEP, q += STEP) {
                for (int i = 0; i < STEP; ++i)
                    if (pf_q < Q) issue_next();
                wait_younger(pf_q - q - STEP);
                consume();
            }
        }

        // ---------------- layers 1..L-1 + likelihood terms of every candidate ----------------
        run_tail(acc0, tseq, tile);
    }
    }

    NPBNN_ESTAMP(4);
    // ---------------- per-workgroup partials (float64, fixed order): waves -> LDS -> global [candidate][value][workgroup] ----
    if (g_partials) {
        constexpr int nvals = (LK == kLikGauss) ? kPartialStride : 1;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if constexpr (LK == kLikGauss) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    A[j].s1[i] = row_sum_f64(A[j].s1[i]);
                    A[j].s2[i] = row_sum_f64(A[j].s2[i]);
                }
            } else {
                A[j].ll = wave_sum_f64(A[j].ll);
            }
        }
        __syncthreads();                                   // every wave is done with its ring: reuse the rings as scratch
        NPBNN_ESTAMP(5);
        double* wsum = reinterpret_cast<double*>(smem + D * IB);    // [candidate][wave][kPartialStride]
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double* ws = wsum + ((size_t)j * wpb + wave) * kPartialStride;
            if constexpr (LK == kLikGauss) {
                if (lane == 0) ws[0] = 0.0;
                if (n == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ws[1 + 4 * kq + i] = A[j].s1[i];
                        ws[1 + NPBNN_MAX_TARGETS + 4 * kq + i] = A[j].s2[i];
                    }
                }
            } else {
                if (lane == 0) ws[0] = A[j].ll;
            }
        }
        __syncthreads();
        for (int item = tid; item < D * nvals; item += blockDim.x) {
            const int j = item / nvals, v = item % nvals;
            double s = 0.0;
            for (int w = 0; w < wpb; ++w) s += wsum[((size_t)j * wpb + w) * kPartialStride + v];
            g_partials[(((size_t)par * kMaxCand + j) * kPartialStride + v) * G + bid] = s;
        }
    }
    NPBNN_ESTAMP(6);
#undef NPBNN_ESTAMP
}

}  // namespace npbnn
