// eval_kernel: fused forward pass + likelihood (the dominant kernel)
// (part of the device code of the npBNN hot path, see npbnn_kernels.hip.h)
#pragma once
#include "npbnn_common.hip.h"
#include "npbnn_chain.hip.h"

namespace npbnn {

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
typedef void (*eval_fn_t)(const EvalParams*, int, int);
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
// The fast Gaussian builds are for <= kFastGaussTargets target columns (every BASELINE regression shape: 1 or 2): all of them
// belong to the lanes of the first quarter (kq = 0) and the moments of columns 2 and 3 of a lane, 8 float64 registers per
// candidate that the general build carries through the whole kernel, are never touched.
constexpr int kFastGaussTargets = 2;

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
// Builds that form what they derive from the thread index where it is used (eval_kernel: top of a pass, run_tail, epilogue) instead
// of letting the compiler hoist it out of the pass and tile loops: the block-structured fast builds, the Gaussian ones for two layer-0
// tiles and the builds for three or more tiles, whose registers are full - they spilled 6-10 such constants per lane around every pass
// (config 5: 5 MB of scratch writes per launch).  The others have room (the one-tile Gaussian builds spill nothing either way) and would
// only pay the extra vector instructions per tile (config 2: +0.3 us per pass; config 4, 21 tiles per wave: 60.2 -> 65.5 us, measured).
#ifndef NPBNN_LAUNDER_TID
#define NPBNN_LAUNDER_TID (FAST && !SPEC && (BLK || (LK == kLikGauss && MT0 >= 2) || MT0 >= 3))
#endif
#ifndef NPBNN_LAUNDER_TAIL
#define NPBNN_LAUNDER_TAIL (FAST && !SPEC && (BLK || (LK == kLikGauss && MT0 >= 2) || MT0 >= 3))
#endif
#ifndef NPBNN_FAST3_WAVES
#define NPBNN_FAST3_WAVES 12
#endif
constexpr int kFastMaxMT0 = 4;      // fast builds exist for layer 0 of up to 16 * kFastMaxMT0 nodes
constexpr int kFastLayers = 3;      // networks of up to this many layers have shape-specialised tails
struct HotParams {
    const int* labels;
    const float* targets;
    const float* inst_w;
    unsigned* confusion;
    float* y_out;
    long long n_rows;
    int use_classw, predict_mode, weight_sets;
    int aux_off_w, aux_off_t;
    int n_layers, C, MTL, lik_kind, k_targets, act_kind, out_kind, final_act, classw_off, pad_masked;
    int slope_off;      // >= 0: every candidate's activation slopes sit in its LDS image at this float offset (general build only)
    int l1_f16;         // NetMeta::l1_f16
    // shape-specialised tails (tile_tail, NLC > 0): image offsets and activation slopes of layers 1 .. kFastLayers-1
    int frag_off[kFastLayers - 1], bias_off[kFastLayers - 1], in_live[kFastLayers - 1];
    float act_prm[kFastLayers - 1];
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
// NLC / ACTC / PLAIN: shape-specialised form for the networks every BASELINE config uses (MTI == 1).  NLC > 0 = the number of
// layers is the compile-time constant NLC (2 or 3): layer 1 then has MT0 k-tiles, later ones one, all of them one output tile,
// and their image offsets come from HotParams (scalar registers) instead of the parameter block; ACTC >= 0 = the activation
// kind is the constant ACTC; PLAIN = categorical log-likelihood only (no confusion counts, predictions, row / class weights,
// final activation; padding outputs masked through their bias).  The arithmetic per value is the generic path's, statement for
// statement - only the control flow around it is resolved at compile time - so both paths give the same bits.
template <int MT0, int MTI, int LK, int D, int DA, int J0, int NLC = 0, int ACTC = -1, bool PLAIN = false, bool F16IMG = false, int GT = 0>
__device__ __forceinline__ void tile_tail(const NetMeta& net, const HotParams& hp, const float* imgs0, int image_floats,
                                          const f32x4 (&acc0_all)[DA][MT0], int lane, int n, int kq, const char* a_slot, float* row_scratch,
                                          long long row, bool row_ok, TileAcc<LK> (&A_all)[DA]) {
    static_assert(J0 + D <= DA, "candidate range");
    constexpr bool primary = J0 == 0;              // statistics and predictions come from the first candidate
    const float* const imgs = imgs0 + (size_t)J0 * image_floats;
    auto A = [&](int j) -> TileAcc<LK>& { return A_all[J0 + j]; };
    constexpr int HT = MT0 > MTI ? MT0 : MTI;      // tiles of the widest activation vector held in registers
    static_assert(NLC == 0 || (MTI == 1 && NLC >= 2 && NLC <= kFastLayers), "shape-specialised tails: 2..kFastLayers layers, narrow later layers");
    static_assert(!PLAIN || (NLC > 0 && LK != kLikGen), "the plain form needs a fixed shape");
    const int n_layers = NLC ? NLC : hp.n_layers;
    const int C = hp.C;
    const int MTL = NLC ? 1 : hp.MTL;
    const int lik_kind = (PLAIN && LK == kLikCat) ? NPBNN_LIK_CATEGORICAL : hp.lik_kind;
    const int k_targets = hp.k_targets;
    const bool need_softmax = LK == kLikCat && (PLAIN || (lik_kind == NPBNN_LIK_CATEGORICAL) || (hp.predict_mode == 2 && hp.out_kind == NPBNN_OUT_SOFTMAX));
    // ---------------- layers 1..L-1 chained through the accumulators ----------------
    f32x4 h[D][HT];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) h[j][mt] = mt < MT0 ? acc0_all[J0 + j][mt < MT0 ? mt : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
    auto layer = [&](int lkt, int lmt, int frag_off, int bias_off, float prm, int hidden, int live_s) {
        if constexpr (ACTC >= 0) {
            act_tiles_all<ACTC>(h, lkt, prm);
        } else if (hp.slope_off >= 0) {           // trainable slopes: each candidate's own, from its image (wave-uniform reads)
#pragma unroll
            for (int j = 0; j < D; ++j) act_live(h[j], lkt, hp.act_kind, imgs[(size_t)j * image_floats + hp.slope_off + hidden]);
        } else {
            act_live_all(h, lkt, hp.act_kind, prm);
        }
        const float* frag = imgs + frag_off + lane * 4;
        const float* bias = imgs + bias_off + 4 * kq;
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
                            if (s < live_s)          // (LayerMeta::in_live: the registers past it hold padding only - wave-uniform)
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
    };
    // layer 1 on fp16-split products (NetMeta::l1_f16; the launch says whether its image is packed that way)
    constexpr bool L1F = F16IMG && MT0 >= 2 && MTI == 1 && (ACTC < 0 || ACTC == NPBNN_ACT_TANH);
    auto layer1_f16 = [&](int frag_off, int bias_off) {
        act_tiles_all<NPBNN_ACT_TANH>(h, MT0, 0.f);
        f32x4 acc[D];
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] = *reinterpret_cast<const f32x4*>(imgs + (size_t)j * image_floats + bias_off + 4 * kq);
#pragma unroll
        for (int q = 0; q < (MT0 + 1) / 2; ++q) {
            f16x8 xh[D], xl[D];
#pragma unroll
            for (int j = 0; j < D; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int t = 2 * q + (e >> 2);
                    const float x = t < MT0 ? h[j][t < MT0 ? t : 0][e & 3] : 0.f;
                    const _Float16 hi = (_Float16)x;
                    xh[j][e] = hi;
                    xl[j][e] = (_Float16)(x - (float)hi);
                }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float* fr = imgs + (size_t)j * image_floats + frag_off + q * 512 + lane * 4;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(fr);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(fr + 256);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[j], acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) h[j][0] = acc[j];
    };
    if constexpr (NLC > 0) {
        if constexpr (L1F) {
            if (hp.l1_f16) layer1_f16(hp.frag_off[0], hp.bias_off[0]);
            else layer(MT0, 1, hp.frag_off[0], hp.bias_off[0], hp.act_prm[0], 0, hp.in_live[0]);
#pragma unroll
            for (int l = 2; l < NLC; ++l) layer(1, 1, hp.frag_off[l - 1], hp.bias_off[l - 1], hp.act_prm[l - 1], l - 1, hp.in_live[l - 1]);
        } else {
#pragma unroll
        for (int l = 1; l < NLC; ++l) layer(l == 1 ? MT0 : 1, 1, hp.frag_off[l - 1], hp.bias_off[l - 1], hp.act_prm[l - 1], l - 1, hp.in_live[l - 1]);
        }
    } else {
        for (int l = 1; l < n_layers; ++l) {
            const LayerMeta& L = net.L[l];
            if constexpr (L1F) {
                if (l == 1 && hp.l1_f16) { layer1_f16(uni(L.frag_off), uni(L.bias_off)); continue; }
            }
            layer(uni(L.kt), uni(L.mt), uni(L.frag_off), uni(L.bias_off), uni(net.act_prm[l - 1]), l - 1, uni(L.in_live));
        }
    }
    if (!PLAIN && hp.final_act) act_live_all(h, MTL, hp.act_kind, uni(net.act_prm[n_layers - 1]));
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
        if (PLAIN || hp.pad_masked) {      // padding outputs sit at kPadLogit (their bias): no per-output predicate
#pragma unroll
            for (int mt = 0; mt < MTI; ++mt)
                if (mt < MTL)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) m[j] = max_nn(m[j], h[j][mt][i]);
#pragma unroll
            for (int j = 0; j < D; ++j) m[j] = quad_max(m[j]);
#pragma unroll
            for (int mt = 0; mt < MTI; ++mt)
                if (mt < MTL)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            const float e = __expf(h[j][mt][i] - m[j]);
                            se[j] = (mt == 0 && i == 0) ? e : se[j] + e;      // (0 + e is not e for the compiler: e could be -0)
                        }
        } else {
#pragma unroll
            for (int mt = 0; mt < MTI; ++mt)
                if (mt < MTL)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (16 * mt + 4 * kq + i < C)
#pragma unroll
                            for (int j = 0; j < D; ++j) m[j] = max_nn(m[j], h[j][mt][i]);
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
        }
#pragma unroll
        for (int j = 0; j < D; ++j) se[j] = quad_sum(se[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) lse[j] = m[j] + log_1_to_n(se[j]);
        if (!PLAIN && hp.confusion && primary) {   // np.argmax: first maximum wins (BNN_lib.py:207); statistics of the first candidate only
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
            if constexpr (!PLAIN) {
                if (hp.inst_w) wgt *= *reinterpret_cast<const float*>(a_slot + hp.aux_off_w + n * 4);
                if (hp.use_classw) wgt *= imgs[hp.classw_off + lab];
            }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float term = 0.f;
                if (own) term += zl[j];
                if (kq == 0) term -= lse[j];
                if constexpr (!PLAIN) term *= wgt;      // (x 1.0f is exact: the plain form gives the same bits)
                A(j).ll += (double)term;
            }
            if (!PLAIN && hp.confusion && primary && kq == 0 && best_i < C) atomicAdd(hp.confusion + lab * C + best_i, 1u);
        }
      }
    } else if constexpr (LK == kLikGen) {
        // (kLikGen builds only: the float64 lgamma / log / exp below would otherwise cost the hot kernels their registers)
        // likelihoods pairing output j with output k+j of the same row (BNN_lib.py:134-143, BNN_lik.py:5-66): the 16
        // outputs of a row meet through LDS; lane (n, kq) then owns target columns j = kq, kq+4, ...; float64 terms
        const float* tg = reinterpret_cast<const float*>(a_slot + hp.aux_off_t);
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
        const float* tg = reinterpret_cast<const float*>(a_slot + hp.aux_off_t);
        constexpr int GI = GT ? GT : (PLAIN ? kFastGaussTargets : 4);          // target columns a lane can own in this build
#pragma unroll
        for (int i = 0; i < GI; ++i) {
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

    if (!PLAIN && hp.predict_mode && row_ok) {       // predictions: of the first candidate, or of every weight set of the launch
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

// software-pipelined layer 0 (the fragments of K-step s+1 are read from LDS while the MFMAs of step s run): builds whose two
// fragment sets fit the register budget of their launch bounds
__host__ __device__ constexpr bool pipelined_l0(int mt0, int mti, bool f16, int d, bool fast_dense = false) {
    // (two candidates: one tile - or, in the fast builds of dense first layers, three and four: 158 registers, nothing spilled, the
    // reference's default [50, 5] +3-5 %; with two tiles the second fragment set does not fit, nor in the other builds of three and more)
    return f16 && mti == 1 && ((d == 3 && mt0 <= 2) || (d == 2 && (mt0 <= 1 || (fast_dense && (mt0 == 3 || mt0 == 4)))) || (d == 1 && mt0 <= 2));
}
// candidates per pass a layer-0 width is built for: three sets of accumulators and tails side by side spill INSIDE the tile loop from
// three output tiles on (tools/check_hot_loop_spills.sh; measured on 100k x 64, hidden [50, 5]: 30.3 us for three candidates
// against 18.1 us for two, chain 65-81 k against 91 k it/s) - plan_launch asks for no more, and no such build is instantiated
__host__ __device__ constexpr int max_cand_for(int mt0) { return mt0 <= 2 ? 3 : 2; }
// waves per workgroup a build is compiled for: more candidates keep more accumulators and weight fragments alive.  The
// three-candidate categorical build of the narrow networks (the chain kernel of config 2) fits 128 VGPRs, i.e. 13 waves:
// with 24-25 tiles per workgroup that is two rounds of tiles per wave instead of three for some.
// The fast three-candidate builds take 12 (three per SIMD, 168 registers): with all three tails unrolled side by side 128 spill.
__host__ __device__ constexpr int max_waves_for(int mt0, int mti, bool f16, int d, int lk, bool fast = false) {
    return mti != 1 ? 8 : d == 1 ? 16 : d == 2 ? (fast ? 12 : 14) : fast ? (NPBNN_FAST3_WAVES) : (lk == kLikCat && pipelined_l0(mt0, mti, f16, d)) ? 13 : 11;
}
// s_waitcnt lgkmcnt(0) as an instruction the compiler's own wait-count bookkeeping sees (the builtin, not inline assembly): with an
// opaque asm statement the compiler added a full wait of its own in front of every unit's first MFMA - behind the LDS reads just
// requested for the NEXT unit, which therefore never ran underneath the MFMAs - and moved the statement itself up between them
#define NPBNN_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)

// The parameter block of a launch is read with scalar loads scattered over the prologue (the network description, the layout, the
// pass descriptor, pointers): each is a round trip of 0.6-0.9 us to the L2 when it misses the scalar cache, and they depend on one
// another (measured, NPBNN_EXP_PROLOGUE_STAMPS: 3.6 us before the first X piece is requested).  This asks for every line of the
// block at once, first thing in the kernel - one round trip - so that the loads that follow hit the scalar cache.  The values are
// not used (the destinations overlap on purpose: only the cache lines matter).
typedef int i32x16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void warm_scalar_cache(const EvalParams* pp) {
    static_assert(sizeof(EvalParams) <= 16 * 64, "one s_load_dwordx16 per 64-byte line");
    i32x16_t a, b, c, d;
    asm volatile(
        "s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\t"
        "s_load_dwordx16 %0, %4, 0x100\n\ts_load_dwordx16 %1, %4, 0x140\n\ts_load_dwordx16 %2, %4, 0x180\n\ts_load_dwordx16 %3, %4, 0x1c0\n\t"
        "s_load_dwordx16 %0, %4, 0x200\n\ts_load_dwordx16 %1, %4, 0x240\n\ts_load_dwordx16 %2, %4, 0x280\n\ts_load_dwordx16 %3, %4, 0x2c0\n\t"
        "s_load_dwordx16 %0, %4, 0x300\n\ts_load_dwordx16 %1, %4, 0x340\n\ts_load_dwordx16 %2, %4, 0x380\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(pp) : "memory");
}

// FAST: the build for launches that want nothing but the likelihood terms of a 2- or 3-layer network with narrow later layers
// (`fast_launch_ok` below says which: every chain pass and plain evaluation of the BASELINE configs).  It holds only the
// shape-specialised tails and none of the scalars the general epilogue keeps alive (statistics, predictions, row weights ...),
// which is what brings the tile loop's scalar registers back under the file's size.  Same arithmetic, same bits.
// BLK (fast builds only): layer 0 has a block structure (NetMeta::l0_begin ...) and the loop skips the (K-unit, tile) pairs
// without weights; the fast builds without it are for dense first layers and carry no test for it.  The general builds always
// honour the structure.
// CHAIN = false: the build for plain evaluations (npbnn_eval: MCMC.__init__, mh_step, statistics) - no pass descriptor, no patch
// lists, no step workgroup, no device-side waits: none of that code, and none of the registers it keeps alive, is in the kernel.
// SPEC = true: the builds NPBNN_SCHED_PERSIST_SERIAL launches - they carry the step workgroup's outcome-speculative rounds
// (spec_rounds) and the evaluating workgroups' side of it (image copy ahead of the flag, accepted entries applied from the
// descriptor).  Builds of their own, because that code beside the tile loop costs the tightest builds registers in the loop.
template <int MT0, int MTI, bool F16, int D, int LK, bool FAST = false, bool BLK = false, bool CHAIN = true, bool SPEC = false>
__global__ void __launch_bounds__(max_waves_for(MT0, MTI, F16, D, LK, FAST) * 64) eval_kernel(const EvalParams* __restrict__ pp, int launch_arg, int n_loop) {
    static_assert(CHAIN || D == 1, "plain builds evaluate one weight set");
    static_assert(!SPEC || (CHAIN && FAST && !BLK), "spec builds: fast chain builds of dense first layers");
    static_assert(!FAST || (MTI == 1 && LK != kLikGen), "fast builds: narrow later layers, categorical or Gaussian likelihood");
    static_assert(!BLK || (FAST && MT0 >= 2), "block-structure builds are fast builds of layers with several output tiles");
    constexpr bool SKIP = !FAST || BLK;          // this build tests which tiles have weights in a K-unit
    // launch index; bit 30: another launch of the batch has been enqueued behind this one (two-stream schedule, sync_step_leave)
    // n_loop > 1: persistent form - this ONE launch stands for the launches launch0 .. launch0 + n_loop - 1 of the flag-ordered
    // overlapped schedule: every workgroup loops over them, ordered by the same device-side flags (no kernel boundary, no second
    // stream); a workgroup that finishes pass L goes straight on to pass L + 1.  Needs every workgroup resident (one per compute
    // unit: grid <= compute units); every wait is bounded (NPBNN_E_SYNC).
    const int launch0 = launch_arg & 0x3fffffff;
    const bool next_enqueued = (launch_arg >> 30) & 1;
    const int launch_end = launch0 + (n_loop > 1 ? n_loop : 1);
    // the parameter block lives in device memory (warm in L2 across the thousands of launches of a chain); a by-value
    // kernel argument of this size costs several microseconds of cold scalar loads per launch
#ifndef NPBNN_EXP_NO_WARM
    warm_scalar_cache(pp);
#endif
    const EvalParams& p = *pp;
    const int bid = uni((int)blockIdx.x);      // (pinned to a scalar register before the first lane-dependent branch)
    unsigned long long* const stamps = uni(p.stamps);
#define NPBNN_ESTAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)bid * 8 + (k)] = wall_clock64(); } while (0)
#ifdef NPBNN_EXP_PROLOGUE_STAMPS      // (diagnostic builds only: finer stamps inside the prologue, a second block behind the per-wave ones)
#define NPBNN_ESTAMPX(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)gridDim.x * 24 + (size_t)bid * 8 + (k)] = wall_clock64(); } while (0)
#else
#define NPBNN_ESTAMPX(k) do { } while (0)
#endif
    constexpr int DEPTH = F16 ? ((kRing - 1) & ~1) : kRing - 1;   // pieces in flight; whole pairs in fp16-split mode
    constexpr bool PIPE = pipelined_l0(MT0, MTI, F16, D, FAST && !BLK && !SPEC) && kRing == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // overlapped chain schedule: the last workgroup decides the previous pass and prepares the next one while the others
    // evaluate this one (chain_step above); passes alternate between two sets of descriptors / patch values / partial sums
    const ChainParams* const chain = CHAIN ? uni(p.chain) : nullptr;
    const int GN = (CHAIN && D > 1) ? uni(p.group_n) : 0;          // group pass: candidate j belongs to chain j, whose step runs in workgroup G + j
    const int G = (int)gridDim.x - (GN ? GN : chain ? 1 : 0);     // workgroups that evaluate
    if (GN && bid >= G) {
        StepShared& sh = *reinterpret_cast<StepShared*>(smem);
        chain_step(*uni(p.group[bid - G].chain), overlapped_plan(launch0), sh);
        return;
    }
    const bool sync = chain && uni(p.sync_mode);        // launches overlap: device flags order them (npbnn_chain.hip.h)
    // the step workgroup is the last one of the launch - or, when launches overlap, the FIRST: it must be resident before any
    // workgroup of the NEXT launch (which waits for it) can take a compute unit
    const int ebid = sync ? bid - 1 : bid;               // index among the evaluating workgroups
    // the share of tiles this workgroup takes in the pass at hand (EvalParams::share_rot: the heavier shares go round in a persistent launch)
    const int share_rot = (CHAIN && sync && !GN) ? uni(p.share_rot) : 0;
    int lbid = ebid;
    for (int i = 0; i < launch0 && share_rot > 0; ++i) { lbid += share_rot; if (lbid >= G) lbid -= G; }
    if (chain && bid == (sync ? 0 : G)) {
        StepShared& sh = *reinterpret_cast<StepShared*>(smem);
        int* const lds_flag = reinterpret_cast<int*>(smem + sizeof(StepShared));
        if (SPEC && uni(p.sync_mode) == 3) {     // NPBNN_SCHED_PERSIST_SERIAL (spec builds): round P prepares pass P + 1 for every outcome of pass P while it is
            // evaluated, then decides it and publishes the pass its outcome selects (spec_rounds)
            SpecShared& sp = *reinterpret_cast<SpecShared*>(smem + ((sizeof(StepShared) + 64 + 15) & ~(size_t)15));
            if (chain->mask) spec_rounds<true>(*chain, launch0, launch_end, G, sh, sp, lds_flag);
            else spec_rounds<false>(*chain, launch0, launch_end, G, sh, sp, lds_flag);
            return;
        }
        for (int launch = launch0; launch < launch_end; ++launch) {
            if (sync && !sync_step_enter(*chain, launch, G, lds_flag)) return;
            chain_step(*chain, overlapped_plan(launch), sh);
            if (sync) sync_step_leave(chain->st, launch + 1, next_enqueued && launch + 1 == launch_end);
            if (launch + 1 < launch_end) {      // persistent form: stop where the evaluating workgroups stop - at the terminal pass
                __syncthreads();
                if (threadIdx.x == 0) {
                    const PassDesc& nx = chain->pass[overlapped_plan(launch).out];
                    *lds_flag = (nx.n_cand == 0 && nx.pad[0] == 1) ? 1 : 0;
                }
                __syncthreads();
                if (*lds_flag) return;
                __syncthreads();
            }
        }
        return;
    }
#ifdef NPBNN_EXP_PROLOGUE_STAMPS
    unsigned long long t_prev_pass0 = 0;
#endif
    // Persistent launch, looking one pass ahead (LOOK builds): a workgroup that has finished the tiles of pass L reads the flag and
    // the descriptor of pass L + 1 while its sums go out and, if that pass is ready and real, requests its weight images at once:
    // the next turn of the loop starts with the descriptor in registers and the images on their way (a flag round trip, a descriptor
    // round trip and the images' latency - 3 us - off every pass's critical path).  ahead_t0 >= 0: pass `launch` was looked at that way.
    constexpr bool LOOK = CHAIN && !SPEC;
    int ahead_t0 = -1;
    int ahead_cnt[D];
#pragma unroll
    for (int j = 0; j < D; ++j) ahead_cnt[j] = 0;
    for (int launch = launch0; launch < launch_end; ++launch, lbid = (lbid + share_rot >= G ? lbid + share_rot - G : lbid + share_rot)) {         // (one pass, but for the persistent form: see n_loop)
    // Every pass reads its parameters afresh, through a pointer the compiler cannot see through: otherwise it hoists the dozens of
    // launch-invariant scalars of the pass out of this loop and keeps them alive across it - far more than the scalar register file
    // holds (70 spilled scalars against 33).
#ifdef NPBNN_EXP_PROLOGUE_STAMPS   // (kept in a register until the pass turns out to be a real one: a persistent launch ends with an empty turn)
    const unsigned long long t_pass0 = stamps ? wall_clock64() : 0;
#else
    NPBNN_ESTAMP(0);          // (start of this pass: of the launch, or of its turn in a persistent launch)
#endif
    const EvalParams* pp_pass = pp;
    asm volatile("" : "+s"(pp_pass));
    // The parameter block does not change while a launch runs (but for the pass descriptors, which a flag-ordered launch reads with
    // vector loads of their own): read through the CONSTANT address space its fields come in by scalar loads that hit the scalar
    // cache.  Through the generic pointer every one of them is a vector load + readfirstlane - the compiler cannot rule out that the
    // kernel's own stores change the block - and their dependent round trips cost a persistent launch 3.3 us at the top of every pass
    // (measured, NPBNN_EXP_PROLOGUE_STAMPS).
    typedef const __attribute__((address_space(4))) EvalParams ConstEvalParams;
    ConstEvalParams& p = *(ConstEvalParams*)pp_pass;
    const EvalParams& p_generic = *pp_pass;
    // (the chain's parameter block sits behind this one in the same buffer and is as constant: its state pointer by a scalar load too -
    // through the generic pointer it was a vector load whose readfirstlane waited for every image and X piece requested before it)
    typedef const __attribute__((address_space(4))) ChainParams ConstChainParams;
    ChainDev* const st_dev = chain ? uni(((ConstChainParams*)chain)->st) : nullptr;
    const SpecState* const spec_dev = (SPEC && chain) ? uni(((ConstChainParams*)chain)->spec) : nullptr;
    unsigned long long* const spec_part = (SPEC && chain) ? uni(((ConstChainParams*)chain)->spec_part) : nullptr;
    const unsigned spec_gen0 = (SPEC && chain) ? (unsigned)uni(((ConstChainParams*)chain)->spec_gen) : 0u;
    // (the thread index through a copy the compiler cannot trace, once at the top of a pass and once behind its tiles: what the prologue
    // and the epilogue derive from it - patch-list addresses, lane tests - is then formed where it is used, every pass, instead of being
    // hoisted out of the pass loop of a persistent launch and kept in registers across the tile loop)
    int tid_pass = threadIdx.x;
    if constexpr (NPBNN_LAUNDER_TID) asm volatile("" : "+v"(tid_pass));
    const int tid = tid_pass;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const auto& net = p.net;
    const NetMeta& net_generic = p_generic.net;          // (for the tails' signature; what they read of it is in `hp`)
    const int wpb = blockDim.x >> 6;
    const int image_floats = uni(net.image_floats);
    const size_t IB = (size_t)image_floats * 4;                     // bytes of one weight image

    HotParams hp;
    hp.n_layers = uni(net.n_layers); hp.C = uni(net.n_out);
    hp.k_targets = uni(net.k_targets); hp.act_kind = uni(net.act_kind);
#pragma unroll
    for (int l = 1; l < kFastLayers; ++l) { hp.frag_off[l - 1] = 0; hp.bias_off[l - 1] = 0; hp.act_prm[l - 1] = 0.f; hp.in_live[l - 1] = 4; }
    if constexpr (FAST) {       // (fast_launch_ok: the host has checked all of this)
        hp.labels = LK == kLikCat ? uni(p.labels) : nullptr;
        hp.targets = LK == kLikGauss ? uni(p.targets) : nullptr;
        hp.inst_w = nullptr; hp.confusion = nullptr; hp.y_out = nullptr;
        hp.n_rows = LK == kLikGauss ? uni(p.n_rows) : 0;        // (categorical: padding rows carry the label -1)
        hp.use_classw = 0; hp.predict_mode = 0; hp.weight_sets = 0;
        hp.MTL = 1; hp.lik_kind = LK == kLikGauss ? NPBNN_LIK_GAUSS : NPBNN_LIK_CATEGORICAL;
        hp.out_kind = 0; hp.final_act = 0; hp.pad_masked = 1; hp.classw_off = -1; hp.slope_off = -1;
        hp.l1_f16 = uni(net.l1_f16);
#pragma unroll
        for (int l = 1; l < kFastLayers; ++l)
            if (l < hp.n_layers) {
                hp.frag_off[l - 1] = uni(net.L[l].frag_off);
                hp.bias_off[l - 1] = uni(net.L[l].bias_off);
                hp.act_prm[l - 1] = uni(net.act_prm[l - 1]);
                hp.in_live[l - 1] = uni(net.L[l].in_live);
            }
    } else {
        hp.labels = uni(p.labels); hp.targets = uni(p.targets); hp.inst_w = uni(p.inst_w); hp.confusion = uni(p.confusion);
        hp.y_out = uni(p.y_out);
        hp.n_rows = uni(p.n_rows); hp.use_classw = uni(p.use_classw); hp.predict_mode = uni(p.predict_mode);
        hp.weight_sets = uni(p.weight_sets);
        hp.MTL = uni(net.L[hp.n_layers - 1].mt); hp.lik_kind = uni(net.lik_kind);
        hp.out_kind = uni(net.out_kind);
        hp.final_act = uni(net.final_act);
        hp.pad_masked = uni(net.pad_masked);
        hp.classw_off = uni(net.classw_off);
        hp.slope_off = uni(p.cand_slopes) != nullptr ? uni(net.slope_off) : -1;
        hp.l1_f16 = uni(net.l1_f16);
    }
    const int k_targets = hp.k_targets;
    const int aux_sz = uni(p.lay.aux_sz), aux_mask = uni(p.lay.aux_slots) - 1;
    hp.aux_off_w = uni(p.lay.off_w);
    hp.aux_off_t = uni(p.lay.off_t);
    const float* const Xg = uni(p.X);
    const int Fp = uni(p.Fp);
    const int n_tiles = uni(p.n_tiles);
    const int M = uni(p.M);
    const int* const g_pos = uni(p.pos);
    const float* const g_pscale = uni(p.pscale);
    double* const g_partials = uni(p.partials);

    NPBNN_ESTAMPX(0);
    char* const ring = smem + D * IB + (size_t)wave * uni(p.lay.wave_lds);
    char* const aux = ring + kRing * 1024;
    float* const row_scratch = reinterpret_cast<float*>(aux + (aux_mask + 1) * aux_sz);   // [16 rows][16 outputs], generic likelihoods

    // ---- stage the weight image of the current state into LDS, once per candidate: lane-linear DMA copies ----
    auto stage_images = [&]() {
        const int n_pieces = (image_floats + 255) >> 8;   // 1-KiB pieces; the last one may be partial (the image is a multiple of 256 B)
        const float* const image = uni(p.image);
        const size_t set_stride = hp.weight_sets ? (size_t)image_floats : 0;
        // pieces outermost, candidates innermost: the copies of a chain pass all read the one committed image - one lane address serves
        // the D copies of a piece - and there is ONE loop: behind a loop of LDS-DMAs the compiler waits for every copy in flight before it
        // reuses the address registers, so a loop per candidate made the images arrive one after the other (three round trips)
        const float* img[D];
#pragma unroll
        for (int j = 0; j < D; ++j) img[j] = GN ? uni(p.group[j].image) : image + j * set_stride;     // (group pass: chain j's own image)
        auto copy_all = [&](auto coherent) {
            for (int i = wave; i < n_pieces; i += wpb)
                if (i * 256 + lane * 4 < image_floats) {    // (an LDS-DMA writes only its active lanes' 16 bytes)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        if constexpr (decltype(coherent)::value) dma16_coherent(img[j] + (size_t)i * 256 + lane * 4, smem + j * IB + (size_t)i * 1024);
                        else dma16(img[j] + (size_t)i * 256 + lane * 4, smem + j * IB + (size_t)i * 1024);
                    }
                }
        };
        if (sync) copy_all(std::true_type{});
        else copy_all(std::false_type{});
    };

    // ---- tile schedule: workgroup b owns tiles b, b+G, b+2G, ...; its m-th tile goes to SIMD m % 4 and, there, to the
    //      waves of that SIMD in turn (wave w sits on SIMD w % 4): the SIMDs of a CU - whose issue ports are what the tile
    //      loop saturates - get the same number of tiles to within one even when the waves do not divide by four ----
    const int KT0 = uni(net.L[0].kt);
    const int first_tile = lbid + G * wave;
    const int simd_waves = (wpb - (wave & 3) + 3) >> 2;      // waves of this workgroup on this wave's SIMD
    const int stride = G * 4 * simd_waves;
    // (counted, not divided, when it is a handful: an integer division runs on the vector unit, and the first vector instruction at
    // the top of a pass waits for every vector memory operation still in flight - the image copies of the look-ahead among them)
    int my_tiles = 0;
    if (first_tile < n_tiles) {
        if (n_tiles - first_tile <= 32 * stride) {
            for (int t = first_tile; t < n_tiles; t += stride) ++my_tiles;
        } else {
            my_tiles = (n_tiles - first_tile + stride - 1) / stride;
        }
    }
    const int Q = my_tiles * KT0;                       // 1-KiB X pieces this wave consumes
    int Dp = PIPE ? kRing : DEPTH;                      // prefetch distance in pieces
    if (Dp > 2 * KT0) Dp = 2 * KT0;                     // at most 3 tiles in flight (aux slots)
    const bool full_depth = (Dp == DEPTH);

    // prefetch cursor: a wave-uniform running source pointer (scalar arithmetic), this lane's constant byte offset inside a piece
    // (row n, 16-byte group kq) and a scalar ring offset
    const float* pf_ptr = Xg + (size_t)first_tile * 16 * (size_t)Fp;
    const unsigned pf_lane = ((unsigned)n * (unsigned)Fp + 4u * (unsigned)kq) * 4u;
    const size_t tile_jump = (size_t)stride * 16 * (size_t)Fp - (size_t)KT0 * 16;
    int pf_q = 0, pf_kt = 0, pf_tile = first_tile, pf_seq = 0, pf_slot = 0;
    auto issue_aux = [&]() {   // row-aux data of a tile travels ahead of its first X piece
        char* a = aux + (pf_seq & aux_mask) * aux_sz;
        const size_t r0 = (size_t)pf_tile * 16;
        if (lane < 16) {
            if (hp.labels) dma4(hp.labels + r0 + lane, a);
            if (hp.inst_w) dma4(hp.inst_w + r0 + lane, a + hp.aux_off_w);
        }
        if (hp.targets) {
            const int total = 16 * k_targets;           // contiguous floats of this tile's targets
            for (int e = 0; e < total; e += 64) {
                const int idx = e + lane;               // (only the lanes with an element take part: an LDS-DMA writes
                if (idx < total)                        //  lane*4 bytes past its base whatever it loaded, and the slot ends at `total`)
                    dma4(hp.targets + r0 * k_targets + idx, a + hp.aux_off_t + e * 4);
            }
        }
    };
    auto issue_next = [&]() {
        if (pf_kt == 0) issue_aux();
        dma16(reinterpret_cast<const float*>(reinterpret_cast<const char*>(pf_ptr) + pf_lane), ring + pf_slot);
        pf_ptr += 16;
        pf_slot = ring_next(pf_slot);
        ++pf_q;
        if (++pf_kt == KT0) { pf_kt = 0; pf_tile += stride; ++pf_seq; pf_ptr += tile_jump; }
    };
    // fp16-split mode: the two pieces of a K-step at once.  They are consecutive 64-byte column groups of the same rows (a tile has an
    // even number of pieces, the ring an even number of slots, and pieces are only ever requested in such pairs): one lane address - the
    // second copy reaches its columns through the instruction's offset, which moves the LDS side by the same 64 bytes, hence the - 64 -
    // and one round of cursor arithmetic.  The scalar unit is one per compute unit: a dozen waves queue for it, and a piece requested on
    // its own costs ~25 scalar instructions (SQ counters: more scalar than vector instructions per tile).
    auto issue_pair = [&]() {
        if (pf_kt == 0) issue_aux();
        const float* const g = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pf_ptr) + pf_lane);
        dma16(g, ring + pf_slot);
        __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)(ring + pf_slot + (1024 - 64)), 16, 64, 0);
        pf_ptr += 32;
        pf_slot += 2048;
        if (pf_slot == kRing * 1024) pf_slot = 0;
        pf_q += 2;
        pf_kt += 2;
        if (pf_kt == KT0) { pf_kt = 0; pf_tile += stride; ++pf_seq; pf_ptr += tile_jump; }
    };
    auto issue_first = [&](int depth) {       // the first `depth` pieces of the wave (fewer when it has fewer)
        if constexpr (F16) {
            // (no loop: in front of a loop of LDS-DMAs - stores and no loads, to its bookkeeping - the compiler drains every memory
            // operation in flight, the image copies among them, when a register it believes pending is used inside)
            static_assert(kRing <= 4, "one statement per pair of ring slots");
            if (0 < depth && pf_q < Q) issue_pair();
            if (2 < depth && pf_q < Q) issue_pair();
        } else {
            for (int i = 0; i < depth && pf_q < Q; ++i) issue_next();
        }
    };
    // NPBNN_SCHED_PERSIST_SERIAL with the next pass prepared ahead (sync_mode 3): nothing the image copy and the first pieces of X
    // depend on changes at the hand-over - the global image is only written right AFTER a flag, and the pass applies the accepted
    // candidate's entries to its LDS copies itself - so both are requested BEFORE the wait for the step workgroup's flag and have
    // landed when it comes.  Everywhere else the image is committed before the flag: copy after it.
    const bool early_copy = SPEC && uni(p.sync_mode) == 3;
    if (early_copy) {
        stage_images();
        issue_first(Dp);
    }

    const bool ahead = LOOK && ahead_t0 >= 0;      // (wave-uniform, the same in every wave: it came out of LDS behind a barrier)
    int early_prepared = 0x7fffffff;
    if (sync && !ahead && !early_copy && threadIdx.x == 0)        // asked for now, looked at where the pass descriptor is needed (below)
        early_prepared = __hip_atomic_load(&st_dev->prepared, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int par = (chain || GN) ? (launch & 1) : 0;
    // ---- which candidates does this pass evaluate?  A chain pass always computes all D weight sets (the step kernel
    //      pads the tail of a batch with unperturbed copies, cnt = 0, whose sums nobody reads): no per-candidate branches ----
    int t0 = 0;
    int cnt[D];
    int t0g[D];                    // group pass: the iteration each chain's candidate belongs to
#pragma unroll
    for (int j = 0; j < D; ++j) { cnt[j] = 0; t0g[j] = 0; }
    const PassDesc* const pass = (CHAIN && uni(p.has_pass)) ? &p_generic.pass_desc[par] : nullptr;
    int pv_slot = par, acc_cnt = 0, acc_slot = -1;    // (sync_mode 3: named by the descriptor, below)
    // (the flag word of the wait: behind everything else in LDS - the image copies may be landing at the front)
    int* const wait_words = reinterpret_cast<int*>(smem + D * IB + (size_t)wpb * uni(p.lay.wave_lds));
    bool entered = true;
    if constexpr (SPEC) {
        if (early_copy) entered = sync_eval_enter_spec(st_dev, spec_dev, p_generic.pass_desc, launch, par, wait_words);
        else if (sync && !ahead) entered = sync_eval_enter(st_dev, launch, wait_words, early_prepared);
    } else {
        if (sync && !ahead) entered = sync_eval_enter(st_dev, launch, wait_words, early_prepared);
    }
    if (!entered) {
        if constexpr (SPEC) NPBNN_WAIT_VMCNT(0);     // (copies requested ahead of the flag must not land in LDS that is no longer ours)
        return;
    }
    asm volatile("" :: "v"(early_prepared));      // (looked at on every path, for the compiler's wait-count bookkeeping: see the end of the tile loop)
    if (GN) {
        int alive = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const PassDesc* const dj = uni(p.group[j].pass) + par;
            const int nc = uni(dj->n_cand);
            alive |= nc;
            cnt[j] = nc > 0 ? uni(dj->cnt[0]) : 0;           // (a chain that has finished its iterations idles: an unpatched copy)
            t0g[j] = uni(dj->t0);
        }
        if (!alive) return;                                   // every chain is through
    } else if (pass && ahead) {
        t0 = ahead_t0;
#pragma unroll
        for (int j = 0; j < D; ++j) cnt[j] = ahead_cnt[j];
    } else if (pass) {
        if (sync) {     // the descriptor was written by a kernel that may still be running: no scalar (cached) loads of it
            // (decision between the passes: the wait brought the descriptor of the outcome the flag named with it)
            const int w = (SPEC && early_copy) ? wait_words[lane & 7]
                                               : __hip_atomic_load(reinterpret_cast<const int*>(pass) + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_readlane(w, 1) == 0) {              // n_cand: an empty pass
                if (threadIdx.x == 0) sync_eval_leave(st_dev, launch);
                // one pass per launch: this launch is over.  Persistent form: over only at the terminal pass (PassDesc.pad[0]); an
                // empty pass before that means the pass in flight may still accept and start the chain's tail again
                if (n_loop <= 1 || __builtin_amdgcn_readlane(w, 5) == 1 || (SPEC && uni(p.sync_mode) == 3)) {
                    if constexpr (SPEC) NPBNN_WAIT_VMCNT(0);
                    return;
                }
                if constexpr (SPEC) NPBNN_WAIT_VMCNT(0);
                __syncthreads();
                continue;
            }
            t0 = __builtin_amdgcn_readlane(w, 0);
#pragma unroll
            for (int j = 0; j < D; ++j) cnt[j] = __builtin_amdgcn_readlane(w, 2 + (j < kMaxCand ? j : 0));
            if (SPEC && uni(p.sync_mode) == 3) {     // the descriptor names what this pass reads (spec_rounds): its own patch values, and - after an
                // accept - the accepted candidate's entries, which the global image may not hold yet
                acc_cnt = __builtin_amdgcn_readlane(w, 6);
                const int named = __builtin_amdgcn_readlane(w, 7);
                pv_slot = named & 0xff;
                acc_slot = (named >> 8) - 1;
            }
        } else {
            // (launches ordered by kernel boundaries: the descriptor was final before this launch began - the step that runs beside this
            // launch writes the OTHER parity's - so it comes in with the rest of the block, by scalar loads that hit the warmed scalar
            // cache; through the generic pointer it was a vector load and a round trip to the L2 ahead of the image copies)
            const auto& cpass = p.pass_desc[par];
            if (uni(cpass.n_cand) == 0) return;                         // the chain batch is finished
            t0 = uni(cpass.t0);
#pragma unroll
            for (int j = 0; j < D; ++j) cnt[j] = uni(cpass.cnt[j < kMaxCand ? j : 0]);
        }
    }

    NPBNN_ESTAMPX(1);
    if (!early_copy) {
        if (!ahead) stage_images();
        NPBNN_ESTAMPX(2);
        issue_first(Dp);
    }
    ahead_t0 = -1;
    NPBNN_ESTAMPX(3);
    const double* const pv = uni(p.pv) + (size_t)pv_slot * kMaxCand * M;
    // overlapped schedule: the step running in this launch raises ChainDev.void_launch when the pass before this one accepts -
    // this pass is then evaluated from a state that no longer exists and nobody will read its sums: polled once per tile
    // (device-scope load, issued before the tile's tail and looked at after it), the waves skip their remaining tiles
    const int* const void_flag = chain ? uni(&st_dev->void_launch) : nullptr;
    int void_seen = -3;
    // ---- candidates = current state + their own touched entries: fetch the first entry per thread now (its latency
    //      hides under the image copy), meet, patch the LDS images, meet again.  The first barrier also waits for this
    //      wave's image pieces and first X pieces (needed next anyway). ----
    int ppos[D];
    double pval[D];
    float psc[D];
    // where candidate j's patch list lives: rows of the pre-drawn positions / scales and the values the step prepared - the
    // chain's own arrays in a group pass (its candidate sits in slot 0 of its pv block), else row t0 + j and slot j of this chain's
    auto patch_src = [&](int j, const int*& pos_row, const float*& psc_row, const double*& pv_row) {
        if (GN) {
            const int Mj = uni(p.group[j].M);
            const float* const sc = uni(p.group[j].pscale);
            pos_row = uni(p.group[j].pos) + (size_t)t0g[j] * Mj;
            psc_row = sc ? sc + (size_t)t0g[j] * Mj : nullptr;
            pv_row = uni(p.group[j].pv) + (size_t)par * kMaxCand * Mj;
        } else {
            pos_row = g_pos + (size_t)(t0 + j) * M;
            psc_row = g_pscale ? g_pscale + (size_t)(t0 + j) * M : nullptr;
            pv_row = pv + (size_t)j * M;
        }
    };
    // Every load of the thread is requested before any of them is waited for (the values are used behind the barrier): through
    // global-address-space pointers - a generic agent-scope load came out with a full wait for everything in flight in front of it,
    // image copies and X pieces included, once per candidate - and with the conversions the compiler would hoist in front of the
    // barrier pinned behind it (below).  Before: 3 us at the top of every pass of a persistent launch, most of it these round trips.
    {
        typedef const __attribute__((address_space(1))) int gint_t;
        typedef const __attribute__((address_space(1))) float gfloat_t;
        typedef const __attribute__((address_space(1))) double gdouble_t;
        const int* pos_rows[D]; const float* psc_rows[D]; const double* pv_rows[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            pos_rows[j] = nullptr; psc_rows[j] = nullptr; pv_rows[j] = nullptr;
            if (pass || GN) patch_src(j, pos_rows[j], psc_rows[j], pv_rows[j]);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            ppos[j] = 0; pval[j] = 0.0; psc[j] = 1.0f;
            if ((pass || GN) && tid < cnt[j]) {
                ppos[j] = ((gint_t*)pos_rows[j])[tid];
                pval[j] = sync ? __hip_atomic_load((gdouble_t*)pv_rows[j] + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((gdouble_t*)pv_rows[j])[tid];
                if (psc_rows[j]) psc[j] = ((gfloat_t*)psc_rows[j])[tid];
            }
        }
    }
    int apos = kSkipPos;
    double aval = 0.0;
    float asc = 1.0f;
    if constexpr (!SPEC) acc_cnt = 0;
    if (acc_cnt > 0 && tid < acc_cnt) {       // the accepted iteration is the one before this pass's first
        const size_t arow = (size_t)(t0 - 1) * M;
        apos = g_pos[arow + tid];
        aval = __hip_atomic_load(uni(p.pv) + (size_t)acc_slot * M + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g_pscale) asc = g_pscale[arow + tid];
    }
#ifdef NPBNN_EXP_PROLOGUE_STAMPS
    if (stamps && threadIdx.x == 0) {
        stamps[(size_t)bid * 8] = t_pass0;
        stamps[(size_t)gridDim.x * 24 + (size_t)bid * 8 + 5] = t_prev_pass0;     // (start of the pass before: the period of a persistent launch)
    }
    t_prev_pass0 = t_pass0;
#endif
    NPBNN_ESTAMP(1);
    __syncthreads();
    NPBNN_ESTAMP(2);
#pragma unroll
    for (int j = 0; j < D; ++j) asm volatile("" : "+v"(ppos[j]), "+v"(pval[j]), "+v"(psc[j]));      // (nothing computed from them in front of the barrier)
    // where the low part of a patched fp16-split entry goes (NetMeta::l0_rows: only first layers of three or more tiles are ever stored
    // with fewer than 16 rows per tile - the builds for one and two tiles keep the constant)
    const int l0_rows_img = MT0 >= 3 ? uni(net.l0_rows) : 16;
    auto lo_halves = [&](int pos) { return (MT0 >= 3 && (pos & kPosCompact)) ? 32 * l0_rows_img : 512; };
    if (acc_cnt > 0) {                        // every candidate starts from the accepted state
        auto put = [&](int pos, double v, float sc) {
            if (pos == kSkipPos) return;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float* imgj = reinterpret_cast<float*>(smem + j * IB);
                if (pos < 0) {
                    _Float16 hi, lo;
                    split_f16((float)(v * (double)sc), hi, lo);
                    _Float16* i16 = reinterpret_cast<_Float16*>(imgj);
                    const int hpos = pos & 0x3fffffff;
                    i16[hpos] = hi;
                    i16[hpos + lo_halves(pos)] = lo;
                } else {
                    imgj[pos] = (float)v;
                }
            }
        };
        if (tid < acc_cnt) put(apos, aval, asc);
        if ((int)blockDim.x < acc_cnt) {
            const size_t arow = (size_t)(t0 - 1) * M;
            for (int e = tid + blockDim.x; e < acc_cnt; e += blockDim.x)
                put(g_pos[arow + e], __hip_atomic_load(uni(p.pv) + (size_t)acc_slot * M + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                    g_pscale ? g_pscale[arow + e] : 1.0f);
        }
        __syncthreads();
    }
    if (pass || GN) {
        auto patch = [&](int j, int pos, double v, float sc) {
            if (pos == 0x7fffffff) return;                  // superseded entry (a later draw of the same position wins)
            float* imgj = reinterpret_cast<float*>(smem + j * IB);
            if (pos < 0) {                               // fp16-split layer-0 entry
                _Float16 hi, lo;
                split_f16((float)(v * (double)sc), hi, lo);
                _Float16* i16 = reinterpret_cast<_Float16*>(imgj);
                const int hpos = pos & 0x3fffffff;
                i16[hpos] = hi;
                i16[hpos + lo_halves(pos)] = lo;
            } else {
                imgj[pos] = (float)v;
            }
        };
        if (hp.slope_off >= 0 && tid < D * kMaxLayers) {      // every candidate's activation slopes into its image copy
            const int j = tid / kMaxLayers, l = tid % kMaxLayers;
            const double* src = uni(p.cand_slopes) + ((size_t)par * kMaxCand + j) * kMaxLayers + l;
            reinterpret_cast<float*>(smem + j * IB)[hp.slope_off + l] =
                (float)(sync ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (tid < cnt[j]) patch(j, ppos[j], pval[j], psc[j]);
            if ((int)blockDim.x < cnt[j]) {                 // (proposals wider than the workgroup)
                const int* pos_row; const float* psc_row; const double* pv_row;
                patch_src(j, pos_row, psc_row, pv_row);
                for (int e = tid + blockDim.x; e < cnt[j]; e += blockDim.x)
                    patch(j, pos_row[e], sync ? __hip_atomic_load(pv_row + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pv_row[e],
                          psc_row ? psc_row[e] : 1.0f);
            }
        }
        __syncthreads();
    }
    NPBNN_ESTAMP(3);

    TileAcc<LK> A[D];
    // Gaussian moments a lane keeps (see kFastGaussTargets; the block-structured fast builds are for ONE target column - block_bnns.py's
    // layout: with two, the three-candidate build spills 100 bytes per lane in its prologue and reloads them in every tail)
    constexpr int GI = FAST ? (BLK ? 1 : kFastGaussTargets) : 4;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if constexpr (LK == kLikGauss) {
#pragma unroll
            for (int i = 0; i < GI; ++i) { A[j].s1[i] = 0.0; A[j].s2[i] = 0.0; }
        } else {
            A[j].ll = 0.0;
        }
    }

    const float* const imgs = reinterpret_cast<const float*>(smem);
    // float offsets inside an image.  Builds for three or more layer-0 tiles read fragment slots of NetMeta::l0_rows rows (16, or fewer
    // when the layer's width is not a multiple of 16: the lanes of the padding rows read the tile's last real row)
    const int l0_rows = (F16 && MT0 >= 3) ? uni(net.l0_rows) : 16;
    const int slot_f = (F16 && MT0 >= 3) ? 32 * l0_rows : 512, lo_f = (F16 && MT0 >= 3) ? 16 * l0_rows : 256;      // floats per slot; low part behind the high part
    const int frag0_off = uni(net.L[0].frag_off) + ((F16 && MT0 >= 3) ? (kq * l0_rows + (n < l0_rows ? n : l0_rows - 1)) * 4 : lane * 4);
    const int bias0_off = uni(net.L[0].bias_off) + 4 * kq;
    // block structure of layer 0 (NetMeta::l0_begin ...): output tile mt has fragments for the K-units ub[mt] .. ue[mt]-1 only, the
    // one of unit u at (uo[mt] + u) fragment slots into the layer-0 block (a slot = 512 floats on the fp16-split path: high and low
    // parts; 256 on the float32 path)
    int ub[MT0], ue[MT0], uo[MT0];
#pragma unroll
    for (int mt = 0; mt < MT0; ++mt) {
        ub[mt] = uni(net.l0_begin[mt]);
        ue[mt] = uni(net.l0_end[mt]);
        uo[mt] = uni(net.l0_base[mt]) - ub[mt];
    }
    auto unit_tiles = [&](int u) {          // bit mt: tile mt has weights in K-unit u (wave-uniform)
        if constexpr (!SKIP) return (1 << MT0) - 1;
        int m = 0;
#pragma unroll
        for (int mt = 0; mt < MT0; ++mt) m |= (u >= ub[mt] && u < ue[mt]) ? (1 << mt) : 0;
        return m;
    };
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
        const char* a_slot = aux + (tseq & aux_mask) * aux_sz;
        // (the lane's coordinates through a copy the compiler cannot trace: every address and constant the tail derives from them is then
        // formed inside the tail - hoisted out of the tile loop, as loop invariants, they are registers alive across layer 0)
        int lane_t = lane;
        if constexpr (NPBNN_LAUNDER_TAIL) asm volatile("" : "+v"(lane_t));
        const int n = lane_t & 15, kq = lane_t >> 4;
        const int lane = lane_t;
        const long long row = (long long)tile * 16 + n;
        // the candidates go through the tail together (their independent chains interleave) while the registers allow
        constexpr int HT = MT0 > MTI ? MT0 : MTI;
        constexpr int DT = (D * HT <= 6 && LK != kLikGen) ? D : 1;
        // (shape tags: NLC layers fixed / activation ACTC / plain - all 0 / -1 / false for the generic tail)
        auto tail = [&](auto nlc, auto actc, auto plain) {
            constexpr int NLC = decltype(nlc)::value, ACTC = decltype(actc)::value;
            constexpr bool PL = decltype(plain)::value;
            if constexpr (DT == D) {
                tile_tail<MT0, MTI, LK, D, D, 0, NLC, ACTC, PL, F16, GI>(net_generic, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
            } else {
                tile_tail<MT0, MTI, LK, 1, D, 0, NLC, ACTC, PL, F16, GI>(net_generic, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
                if constexpr (D > 1) tile_tail<MT0, MTI, LK, 1, D, 1, NLC, ACTC, PL, F16, GI>(net_generic, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
                if constexpr (D > 2) tile_tail<MT0, MTI, LK, 1, D, 2, NLC, ACTC, PL, F16, GI>(net_generic, hp, imgs, image_floats, acc0, lane, n, kq, a_slot, row_scratch, row, row < hp.n_rows, A);
                static_assert(D <= 3, "add a call per candidate");
            }
        };
        using std::integral_constant;
        if constexpr (FAST) {      // wave-uniform branches: depth and activation of the network
            auto with_act = [&](auto nlc) {
                switch (hp.act_kind) {
                    case NPBNN_ACT_RELU: tail(nlc, integral_constant<int, NPBNN_ACT_RELU>{}, std::true_type{}); break;
                    case NPBNN_ACT_LEAKY: tail(nlc, integral_constant<int, NPBNN_ACT_LEAKY>{}, std::true_type{}); break;
                    case NPBNN_ACT_SWISH: tail(nlc, integral_constant<int, NPBNN_ACT_SWISH>{}, std::true_type{}); break;
                    default: tail(nlc, integral_constant<int, NPBNN_ACT_TANH>{}, std::true_type{}); break;
                }
            };
            static_assert(kFastLayers == 3, "one branch per specialised depth");
            if (hp.n_layers == 3) with_act(integral_constant<int, 3>{});
            else with_act(integral_constant<int, 2>{});
        } else {
            tail(integral_constant<int, 0>{}, integral_constant<int, -1>{}, std::false_type{});
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
            // LDS addresses as 32-bit byte offsets from smem, the per-lane parts formed once: a fragment read then costs one vector
            // add (lane part + scalar part) - as generic pointers with a per-lane select the same reads took a dozen vector
            // instructions per K-step, half of the non-matrix vector work of the tile loop
            static_assert(kRing == 4, "a K-step is two ring slots: the steps alternate between slots (0, 1) and (2, 3)");
            const unsigned x_lane = (unsigned)(D * IB) + (unsigned)wave * (unsigned)uni(p.lay.wave_lds)
                                    + (unsigned)((kq >> 1) * 1024 + ((2 * (kq & 1)) * 16 + n) * 16);
            unsigned w_lane[D];
#pragma unroll
            for (int j = 0; j < D; ++j) w_lane[j] = (unsigned)(j * IB) + (unsigned)frag0_off * 4u;
            auto load_x = [&](XFrag& x) {
                const char* px = smem + (x_lane + (unsigned)ld_slot);
                x.xh = *reinterpret_cast<const f16x8*>(px);
                x.xl = *reinterpret_cast<const f16x8*>(px + 256);
                ld_slot ^= 2048;
            };
            auto load_w = [&](WFrag& w, int kstep, int j) {
                const unsigned slot_b = (unsigned)slot_f * 4u, lo_b = (unsigned)lo_f * 4u;      // (2048 / 1024 bytes but for NetMeta::l0_rows < 16)
                const unsigned fr = w_lane[j] + (unsigned)kstep * slot_b;
                const int live = unit_tiles(kstep);
#pragma unroll
                for (int mt = 0; mt < MT0; ++mt)
                    if (live & (1 << mt)) {
                        const char* pw = smem + (fr + (unsigned)uo[mt] * slot_b);
                        w.wh[mt] = *reinterpret_cast<const f16x8*>(pw);
                        w.wl[mt] = *reinterpret_cast<const f16x8*>(pw + lo_b);
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
                const int live = unit_tiles(ks);          // tiles with weights in this K-step (all of them in a dense layer)
                bool issued = false;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const WFrag& wc = Wb[(PAR * D + j) & 1];
                    WFrag& wn = Wb[(PAR * D + j + 1) & 1];
                    NPBNN_WAIT_LGKM0();                   // this unit's fragments are complete
                    if (j == 0 && pf_q < Q) {             // the x fragments of step s are in registers: refill its slots (step s+2)
                        issue_pair();
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
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.wh[mt], Xb[PAR].xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.wl[mt], Xb[PAR].xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc.wh[mt], Xb[PAR].xl, acc0[j][mt], 0, 0, 0);
#else
                    acc0[j][0][0] += (float)wc.wh[0][0] + (float)wc.wl[MT0 - 1][7] + (float)Xb[PAR].xh[0] + (float)Xb[PAR].xl[7];
#endif
                    __builtin_amdgcn_sched_barrier(0);    // ... and the next unit's wait behind them
                }
                ++s;
                ks = ks_next;
            };
            int tile = first_tile;
            unsigned long long tail_ticks = 0;
            for (int tseq = 0; tseq < my_tiles; ++tseq, tile += stride) {
                if (void_flag && void_seen == launch) break;
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
                if (void_flag) void_seen = __hip_atomic_load(void_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (void_flag && void_seen == launch) break;
        // ---------------- layer 0: H0^T = W0 . X^T, K streamed from the ring, every candidate on the same X piece ----------------
        f32x4 acc0[D][MT0];
        load_bias0(acc0);
        int fr_off = frag0_off;            // + K-unit * slot size
        int unit = 0;
        auto consume = [&]() {
            const int live = unit_tiles(unit);
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
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) {
                            wh[mt] = *reinterpret_cast<const f16x8*>(fr + uo[mt] * slot_f);
                            wl[mt] = *reinterpret_cast<const f16x8*>(fr + uo[mt] * slot_f + lo_f);
                        }
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt], xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[mt], xh, acc0[j][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[mt], xl, acc0[j][mt], 0, 0, 0);
                }
                fr_off += slot_f;
                ++unit;
            } else {
                const f32x4 x = *reinterpret_cast<const f32x4*>(ring + cs_slot + lane * 16);
                cs_slot = ring_next(cs_slot);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float* fr = imgs + (size_t)j * image_floats + fr_off;
                    f32x4 a[MT0];
#pragma unroll
                    for (int mt = 0; mt < MT0; ++mt)
                        if (live & (1 << mt)) a[mt] = *reinterpret_cast<const f32x4*>(fr + uo[mt] * 256);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int mt = 0; mt < MT0; ++mt)
                            if (live & (1 << mt)) acc0[j][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s], x[s], acc0[j][mt], 0, 0, 0);
                }
                fr_off += 256;
                ++unit;
            }
        };
        constexpr int STEP = F16 ? 2 : 1;               // pieces per consume()
        if (full_depth) {
            // steady part: every step consumed is replaced by one issued -> exactly DEPTH younger pieces in flight;
            // once the wave's last piece has been issued, drain once and consume what is left without waiting
            int n_issue = Q - pf_q;
            if (n_issue > KT0) n_issue = KT0;
            for (int kt = 0; kt < n_issue; kt += STEP) {
                if constexpr (F16) issue_pair();        // targets the slot(s) consumed one step ago
                else issue_next();
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
                if (pf_q < Q) {
                    if constexpr (F16) issue_pair();
                    else issue_next();
                }
                wait_younger(pf_q - q - STEP);
                consume();
            }
        }

        // ---------------- layers 1..L-1 + likelihood terms of every candidate ----------------
        if (void_flag) void_seen = __hip_atomic_load(void_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run_tail(acc0, tseq, tile);
    }
    }
    // (a wave that left early still has X pieces on their way into its ring.  As an instruction the compiler's wait-count bookkeeping
    // sees, on every path: a load whose value is looked at on some paths only - the void flag of the last tile - stays "pending" for it,
    // and the next write to that register, a pass later and behind freshly requested copies, is made to wait for everything in flight)
    __builtin_amdgcn_s_waitcnt(0x0f70);       // s_waitcnt vmcnt(0)
    const int tid_tiles = tid;
    {
    int tid_epi = tid_tiles;
    if constexpr (NPBNN_LAUNDER_TID) {          // (see the top of the pass)
        tid_epi = threadIdx.x;
        asm volatile("" : "+v"(tid_epi));
    }
    const int tid = tid_epi;
    const int lane = tid & 63;
    const int n = lane & 15, kq = lane >> 4;

    NPBNN_ESTAMP(4);
    if (stamps && lane == 0) stamps[(size_t)gridDim.x * 8 + (size_t)bid * 16 + wave] = wall_clock64();   // every wave: tiles done
    const bool look = LOOK && sync && n_loop > 1 && launch + 1 < launch_end && !GN && g_partials != nullptr;
    int* const look_words = reinterpret_cast<int*>(smem + D * IB + (size_t)wpb * uni(p.lay.wave_lds));       // (the flag word's 64 bytes)
    // ---------------- per-workgroup partials (float64, fixed order): waves -> LDS -> global [candidate][value][workgroup] ----
    if (g_partials || GN) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if constexpr (LK == kLikGauss) {
#pragma unroll
                for (int i = 0; i < GI; ++i) {
                    A[j].s1[i] = row_sum_f64(A[j].s1[i]);
                    A[j].s2[i] = row_sum_f64(A[j].s2[i]);
                }
            } else {
                A[j].ll = wave_sum_f64(A[j].ll);
            }
        }
        // (looking ahead: the flag of the pass after this one, asked for before the barrier, read behind it)
        int nx_prepared = -1;
        if (look && wave == 0)
            nx_prepared = __hip_atomic_load(&st_dev->prepared, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                   // every wave is done with its ring: reuse the rings as scratch
        NPBNN_ESTAMP(5);
        int nx_desc = 0;
        bool nx_ready = false;
        if (look && wave == 0) {
            nx_ready = uni(nx_prepared) >= launch + 1;      // (ready: its descriptor is final - asked for only now, after the flag was seen)
            if (nx_ready)
                nx_desc = __hip_atomic_load(reinterpret_cast<const int*>(&p_generic.pass_desc[(launch + 1) & 1]) + (lane & 7), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
        }
        double* wsum = reinterpret_cast<double*>(smem + D * IB);    // [candidate][wave][kPartialStride]
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double* ws = wsum + ((size_t)j * wpb + wave) * kPartialStride;
            if constexpr (LK == kLikGauss) {
                if (lane == 0) ws[0] = 0.0;
                if (n == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ws[1 + 4 * kq + i] = i < GI ? A[j].s1[i < GI ? i : 0] : 0.0;
                        ws[1 + NPBNN_MAX_TARGETS + 4 * kq + i] = i < GI ? A[j].s2[i < GI ? i : 0] : 0.0;
                    }
                }
            } else {
                if (lane == 0) ws[0] = A[j].ll;
            }
        }
        if (look && wave == 0 && lane < 8)                 // n_cand, t0, cnt ... of the pass ahead; n_cand = 0 when it is not to be started early
            look_words[lane] = nx_ready ? nx_desc : 0;
        asm volatile("" :: "v"(nx_prepared), "v"(nx_desc));      // (both loads are through for the compiler's bookkeeping on every path, see above)
        __syncthreads();
        // (the LAST wave adds them up and writes them: in a flag-ordered launch it is also the one that reports the workgroup done,
        // after waiting for these very stores)
        constexpr int nvals = (LK == kLikGauss) ? kPartialStride : 1;
        for (int item = tid - ((int)blockDim.x - 64); item >= 0 && item < D * nvals; item += 64) {
            const int j = item / nvals, v = item % nvals;
            // (only the values in use leave the workgroup - 1 + 2k of the Gaussian record's 33, the ones every reader asks for: partial_value_index)
            if constexpr (LK == kLikGauss) { if (v != 0 && ((v - 1) & (NPBNN_MAX_TARGETS - 1)) >= p_generic.net.k_targets) continue; }
            double s = 0.0;
            for (int w = 0; w < wpb; ++w) s += wsum[((size_t)j * wpb + w) * kPartialStride + v];
            // (group pass: candidate j is slot 0 of chain j's own partial block)
            double* const dst = GN ? p.group[j].partials + (((size_t)par * kMaxCand) * kPartialStride + v) * G + lbid
                                   : g_partials + (((size_t)par * kMaxCand + j) * kPartialStride + v) * G + lbid;
            if constexpr (SPEC) {
                // (decision between the passes: the sums travel as tagged word pairs - nothing to wait for, nothing to report, the step
                // reads the tags with the values: spec_part_store)
                if (early_copy) { spec_part_store(spec_part + spec_part_index(par, j, v, lbid), s, spec_gen0 + (unsigned)launch + 1u); continue; }
            }
            if (sync) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (write-through: no fence at the end)
            else *dst = s;
        }
    }
    NPBNN_ESTAMP(6);
    if (sync) {                                  // this workgroup's sums are out: tell the step of the next launch
        __syncthreads();
        // (by the LAST wave: it waits for the stores to be acknowledged, a round trip that wave 0 - which reads the next pass's flag
        // and descriptor for everybody - does not have to sit through)
        if (threadIdx.x == blockDim.x - 64 && !(SPEC && early_copy)) sync_eval_leave(st_dev, launch);
        if constexpr (LOOK) {
            // (look_words were written before the barrier two above: every wave reads the same descriptor)
            if (look && look_words[1] > 0) {     // word 1: n_cand
                ahead_t0 = uni(look_words[0]);
#pragma unroll
                for (int j = 0; j < D; ++j) ahead_cnt[j] = uni(look_words[2 + (j < kMaxCand ? j : 0)]);
                stage_images();                 // (every wave is past its tiles: the image slots are free; the step of the pass ahead does
                                                //  not touch the global image before every workgroup - this one too - has reported done)
            }
        }
    }
    }
    NPBNN_ESTAMPX(4);
    }       // (next pass of the persistent form)
#undef NPBNN_ESTAMP
}

}  // namespace npbnn
