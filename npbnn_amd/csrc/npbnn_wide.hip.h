// The weight-streamed path: networks whose weight image does not fit a compute unit's LDS (wide layers, many features)
// (part of the device code of the npBNN hot path, see npbnn_kernels.hip.h)
//
// The resident path (npbnn_eval.hip.h) keeps the whole network in LDS and chains the layers through the accumulators of one
// wavefront.  That stops where the image stops fitting: the reference's own default [50, 5] (np_bnn/BNN_env.py:20) above ~650
// features, any layer wider than 128 nodes.  np.dot (MatrixMultiplicationD, np_bnn/BNN_lib.py:154-162) has no such limit, so this
// path has none either: every layer is a tiled matrix product whose operands BOTH stream through LDS -
//
//   wide_gemm_kernel   out[N x H] = act(in[N x K] . W^T + b): a workgroup owns a (16 RT WR) x (16 CT WC) block of the output; per
//                      K-unit of 32 columns the rows' pieces of `in` and the fragments of W arrive by LDS-DMA
//                      (global_load_lds_dwordx4) into a ring of stages, a counted s_waitcnt + one barrier per unit hands a stage
//                      over, every wave multiplies its RT x CT tiles out of LDS.  Layer 0 contracts the feature matrix: fp16-split
//                      operands (3 x v_mfma_f32_16x16x32_f16 per tile and unit, float32 accumulate - the resident path's
//                      arithmetic) or exact float32 (v_mfma_f32_16x16x4_f32); later layers are float32.  Hidden activations go to a
//                      float32 buffer in HBM (written once, read once by the next layer's launch - a few per cent of X's bytes for
//                      the shapes this path exists for).
//   wide_lik_kernel    the last layer's values -> per-row likelihood terms (the resident epilogue's arithmetic: float32 per row,
//                      float64 sums, fixed order), confusion counts, predictions; one partial record per workgroup, summed by
//                      finalize_kernel / chain_step exactly like the resident path's.
//   wide_pack_kernel   float64 packed weights -> the streamed image (layer blocks in K-unit-major fragment order).
//   A device chain's candidate = the committed image with the pending proposal's entries patched in - what the resident path does
//   to its LDS copy is done here to a second image in HBM, by the chain step itself (ChainParams::cand_image: patched when a
//   proposal is prepared, put back when it is rejected).
#pragma once
#include "npbnn_common.hip.h"
#include "npbnn_pack.hip.h"
#include "npbnn_chain.hip.h"

namespace npbnn {

struct WideLayer {
    long long frag_off;   // float offset of the fragment block in the image
    long long bias_off;   // float offset of the 16 * mt bias floats
    int in_dim, out_dim, has_bias, w_off;
    int mt;               // 16-unit output tiles
    int units;            // K-units of 32 input columns
    int f16;              // fp16-split fragments (layer 0 only)
    int pad_;
};
struct WideMeta {
    int n_layers, n_out;
    long long image_floats;
    long long classw_off;     // class weights (16 * mt of the last layer floats), or -1
    WideLayer L[kMaxLayers];
};

// Fragment blocks (16-byte entries, one per lane; o = 16 mt + (lane & 15)):
//   fp16-split layer 0 : entry (((u * MT + mt) * 2 + part) * 64 + lane) = part (0 high, 1 low) of W[o][32 u + 8 (lane >> 4) + 0..7] * w_scale
//   float32            : entry (((2 u + h) * MT + mt) * 64 + lane) = W[o][32 u + 16 h + 4 (lane >> 4) + 0..3]
// - per K-unit u the 2 KiB of every output tile lie side by side: a stage of the ring is one contiguous run per tile.
__host__ __device__ inline long long wide_frag_items(const WideLayer& L) { return (long long)L.units * L.mt * 128; }
__host__ __device__ inline long long wide_item_count(const WideMeta& m) {
    long long total = m.classw_off >= 0 ? 16 * m.L[m.n_layers - 1].mt : 0;
    for (int l = 0; l < m.n_layers; ++l) total += wide_frag_items(m.L[l]) + 16 * m.L[l].mt;
    return total;
}

#ifdef NPBNN_KERNELS_WIDE
__global__ void __launch_bounds__(256) wide_pack_kernel(const double* __restrict__ w, const double* __restrict__ col_override,
                                                        const double* __restrict__ class_w, float* __restrict__ image, const WideMeta m,
                                                        const float* __restrict__ w_scale, int* overflow) {
    long long piece = (long long)blockIdx.x * 256 + threadIdx.x;
    for (int l = 0; l < m.n_layers; ++l) {
        const WideLayer& L = m.L[l];
        const long long n_items = wide_frag_items(L);
        if (piece < n_items) {
            const int lane = (int)(piece & 63);
            const long long tile = piece >> 6;
            const int ld = L.in_dim + L.has_bias;
            if (L.f16) {
                const int part = (int)(tile & 1);
                const long long rest = tile >> 1;
                const int mt = (int)(rest % L.mt), u = (int)(rest / L.mt);
                const int o = 16 * mt + (lane & 15), c0 = 32 * u + 8 * (lane >> 4);
                f16x8 v;
                for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
                if (o < L.out_dim) {
                    const double* row = w + L.w_off + (long long)o * ld + L.has_bias;
                    for (int j = 0; j < 8; ++j) {
                        const int c = c0 + j;
                        if (c < L.in_dim) {
                            const bool overridden = (l == 0 && col_override != nullptr && !isnan(col_override[c]));
                            const float wv = overridden ? 0.f : (float)(row[c] * (double)w_scale[c]);
                            if (overflow && !(fabsf(wv) <= kF16Safe)) atomicOr(overflow, kFlagF16Range);
                            _Float16 hi, lo;
                            split_f16(wv, hi, lo);
                            v[j] = part ? lo : hi;
                        }
                    }
                }
                *reinterpret_cast<f16x8*>(image + L.frag_off + piece * 4) = v;
            } else {
                const int mt = (int)(tile % L.mt);
                const int kt = (int)(tile / L.mt);          // 16-column step: 2 u + h
                const int o = 16 * mt + (lane & 15), c0 = 16 * kt + 4 * (lane >> 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (o < L.out_dim) {
                    const double* row = w + L.w_off + (long long)o * ld + L.has_bias;
                    for (int s = 0; s < 4; ++s) {
                        const int c = c0 + s;
                        if (c < L.in_dim) {
                            const bool overridden = (l == 0 && col_override != nullptr && !isnan(col_override[c]));
                            v[s] = overridden ? 0.f : (float)row[c];
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(image + L.frag_off + piece * 4) = v;
            }
            return;
        }
        piece -= n_items;
    }
    for (int l = 0; l < m.n_layers; ++l) {
        const WideLayer& L = m.L[l];
        const int nb = 16 * L.mt;
        if (piece < nb) {
            const int o = (int)piece;
            double b = 0.0;
            if (o < L.out_dim) {
                const int ld = L.in_dim + L.has_bias;
                const double* row = w + L.w_off + (long long)o * ld;
                if (L.has_bias) b = row[0];
                if (l == 0 && col_override != nullptr) {         // (data_transform_obj, np_bnn/BNN_env.py:14-17: a constant column moves into the bias)
                    for (int c = 0; c < L.in_dim; ++c) {
                        const double ov = col_override[c];
                        if (!isnan(ov)) b += ov * row[L.has_bias + c];
                    }
                }
            }
            image[L.bias_off + piece] = (float)b;
            return;
        }
        piece -= nb;
    }
    if (m.classw_off >= 0 && piece < 16 * m.L[m.n_layers - 1].mt)
        image[m.classw_off + piece] = (class_w != nullptr && piece < m.n_out) ? (float)class_w[piece] : 1.0f;
}
#endif  // NPBNN_KERNELS_WIDE

// ------------------------------------------------------------------------------------------------
// helpers shared by the kernels below
// ------------------------------------------------------------------------------------------------
// The narrow end of a network - layers of <= 128 nodes behind an input of <= 256 (the [64] of [256, 64], the [5] of [50, 5], every
// output layer of a few classes): their weights (a few tens of KiB) sit in LDS, a wave takes a 16-row tile of the activations and
// chains the layers through its accumulators - the resident path's scheme: the accumulator of one layer is the B operand of the next
// (v_mfma_f32_16x16x4_f32, exact float32), bias = initial accumulator, activation elementwise.
constexpr int kTailIn = 16;       // 16-unit tiles of the activations it reads at most
constexpr int kTailOut = 8;       // ... and of any layer it computes
struct WideTailDesc {
    int n_layers;             // layers computed (0: none)
    int last_is_output;       // the last of them is the network's last layer: no activation behind it
    int act_kind;
    int pad_;
    const float* image;       // weight image (global)
    long long frag_off[kMaxLayers], bias_off[kMaxLayers];     // of each layer, in the image
    int mt[kMaxLayers];
    int lds_frag[kMaxLayers], lds_bias[kMaxLayers];           // float offsets of their copies in LDS
    int frag_floats[kMaxLayers];
    float act_prm[kMaxLayers];
    const double* act_prm_dev;    // device chain with trainable slopes: the candidate's slopes of these layers, or nullptr
};
// the layers' fragments and biases into LDS (every thread of the workgroup; a barrier follows at the caller)
__device__ __forceinline__ void wide_tail_stage(const WideTailDesc& t, float* lds, int tid, int n_threads, const float* image = nullptr) {
    if (image == nullptr) image = t.image;
    for (int l = 0; l < t.n_layers; ++l) {
        const f32x4* gf = reinterpret_cast<const f32x4*>(image + t.frag_off[l]);
        f32x4* lf = reinterpret_cast<f32x4*>(lds + t.lds_frag[l]);
        for (int i = tid; i < t.frag_floats[l] / 4; i += n_threads) lf[i] = gf[i];
        const f32x4* gb = reinterpret_cast<const f32x4*>(image + t.bias_off[l]);
        f32x4* lb = reinterpret_cast<f32x4*>(lds + t.lds_bias[l]);
        for (int i = tid; i < 4 * t.mt[l]; i += n_threads) lb[i] = gb[i];
    }
}
// h[0 .. kt-1]: a 16-row tile of activations (lane (n, kq) holds the units 4 kq .. 4 kq + 3 of every 16-unit tile for row n); on
// return h[0 .. result-1] holds the last computed layer's values
__device__ __forceinline__ int wide_tail_layers(const WideTailDesc& t, const float* lds, f32x4 (&h)[kTailIn], int kt, int lane, int kq) {
    for (int l = 0; l < t.n_layers; ++l) {
        const int mt_l = __builtin_amdgcn_readfirstlane(t.mt[l]);
        const float* frag = lds + t.lds_frag[l] + lane * 4;
        const float* bias = lds + t.lds_bias[l] + 4 * kq;
        f32x4 acc[kTailOut];
#pragma unroll
        for (int mt = 0; mt < kTailOut; ++mt) {
            acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mt < mt_l) {
                acc[mt] = *reinterpret_cast<const f32x4*>(bias + 16 * mt);
#pragma unroll
                for (int ct = 0; ct < kTailIn; ++ct)
                    if (ct < kt) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(frag + (size_t)(ct * mt_l + mt) * 256);
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], h[ct][s], acc[mt], 0, 0, 0);
                    }
            }
        }
        const bool last = l + 1 == t.n_layers;
        if (!(last && t.last_is_output)) {
            const float prm = t.act_prm_dev != nullptr ? (float)t.act_prm_dev[l] : t.act_prm[l];
#pragma unroll
            for (int mt = 0; mt < kTailOut; ++mt)
                if (mt < mt_l)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[mt][i] = act_apply(acc[mt][i], t.act_kind, prm);
        }
#pragma unroll
        for (int ct = 0; ct < kTailIn; ++ct) h[ct] = ct < kTailOut ? acc[ct < kTailOut ? ct : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
        kt = mt_l;
    }
    return kt;
}

// Likelihood terms / statistics / predictions of ONE data row from the last layer's values z[0 .. C-1] (no activation applied): the
// resident epilogue's arithmetic - float32 per row, float64 sums.
struct WideRowAcc {
    double ll;
    double s1[NPBNN_MAX_TARGETS], s2[NPBNN_MAX_TARGETS];
};
__device__ __forceinline__ void wide_row_terms(const EvalParams& p, const float* z, long long row, const float* image, long long classw_off,
                                               float final_prm, WideRowAcc& A) {
    const NetMeta& net = p.net;
    const int C = net.n_out, lik_kind = net.lik_kind, k = net.k_targets;
    const int act_kind = net.act_kind;
    auto val = [&](int o) -> float {
        const float v = z[o];
        return net.final_act ? act_apply(v, act_kind, final_prm) : v;
    };
    const bool need_softmax = lik_kind == NPBNN_LIK_CATEGORICAL || (p.predict_mode == 2 && net.out_kind == NPBNN_OUT_SOFTMAX);
    float lse = 0.f;
    int best = 0;
    if (need_softmax) {
        float m = -INFINITY;
        for (int o = 0; o < C; ++o) {
            const float v = val(o);
            if (v > m) { m = v; best = o; }          // np.argmax: the first maximum (BNN_lib.py:207)
        }
        float se = 0.f;
        for (int o = 0; o < C; ++o) se += __expf(val(o) - m);
        lse = m + log_1_to_n(se);
    }
    if (lik_kind == NPBNN_LIK_CATEGORICAL && p.labels != nullptr) {
        const int lab = p.labels[row];
        if (lab >= 0) {
            float wgt = 1.f;
            if (p.inst_w) wgt *= p.inst_w[row];
            if (p.use_classw && classw_off >= 0) wgt *= image[classw_off + lab];
            float term = (lab < C ? val(lab) : 0.f) - lse;
            term *= wgt;
            A.ll += (double)term;
            if (p.confusion) atomicAdd(p.confusion + (size_t)lab * C + best, 1u);
        }
    } else if (lik_kind == NPBNN_LIK_GAUSS && p.targets != nullptr) {
#pragma unroll
        for (int j = 0; j < NPBNN_MAX_TARGETS; ++j)          // (unrolled under a predicate: a run-time index would put the sums in scratch)
            if (j < k) {
                const float r = p.targets[row * k + j] - val(j);
                A.s1[j] += (double)r;
                A.s2[j] += (double)r * (double)r;
            }
    } else if (lik_needs_row_scratch(lik_kind) && p.targets != nullptr) {
        // float64 row-wise likelihoods (BNN_lib.py:134-143, BNN_lik.py:5-66), as the resident path's generic epilogue has them
        double term = 0.0;
        for (int j = 0; j < k; ++j) {
            const double y = (double)p.targets[row * k + j];
            if (lik_kind == NPBNN_LIK_GAUSS_PRED_SIGMA) {
                const double mu = (double)val(j);
                const double zs = (double)val(k + j);
                const double sg = fmax(zs, 0.0) + log1p(exp(-fabs(zs)));
                const double r = (y - mu) / sg;
                term += -0.9189385332046727418 - log(sg) - 0.5 * r * r;
            } else if (lik_kind == NPBNN_LIK_POISSON) {
                if (j == 0) {
                    const double eta = (double)val(0);
                    term += y * eta - exp(eta) - lgamma(y + 1.0);
                }
            } else {
                const bool one_col = lik_kind != NPBNN_LIK_NEGBIN2D;
                if (one_col && j > 0) continue;
                const int jp = one_col ? 1 : k + j;
                const double e0 = (double)val(j), e1 = (double)val(jp);
                double mean, pr;
                if (lik_kind == NPBNN_LIK_NEGBIN_BASE10) {
                    mean = exp(2.302585092994046 * e0);
                    pr = 1.0 / (1.0 + exp(-2.302585092994046 * e1));
                } else {
                    mean = exp(e0);
                    pr = 1.0 / (1.0 + exp(-e1));
                }
                const double nn = pr * mean / (1.0 - pr);
                term += lgamma(y + nn) - lgamma(y + 1.0) - lgamma(nn) + nn * log(pr) + y * log1p(-pr);
            }
        }
        A.ll += term;
    }
    if (p.predict_mode && p.y_out != nullptr) {
        float* yo = p.y_out + row * C;
        for (int o = 0; o < C; ++o) {
            float v = val(o);
            if (p.predict_mode == 2) {
                if (net.out_kind == NPBNN_OUT_SOFTMAX) v = __expf(v - lse);
                else if (net.out_kind == NPBNN_OUT_SOFTPLUS_HALF && o >= C / 2) v = softplus_f(v);
            }
            yo[o] = v;
        }
    }
}
// The same for the product's fused end (WideGemmArgs::fuse), without what would cost that kernel its registers: the categorical and the
// Gaussian likelihood (up to kFuseTargets target columns) and predictions; the float64 row-wise likelihoods stay with wide_lik_kernel.
constexpr int kFuseTargets = 4;
struct WideRowAccLean {
    double ll;
    double s1[kFuseTargets], s2[kFuseTargets];
};
struct WideRowAux {           // what a row's terms read beside its values, requested ahead of them (a dependent miss per tile otherwise)
    int lab;
    float inst_w;
    float tg[kFuseTargets];
};
__device__ __forceinline__ WideRowAux wide_row_aux(const EvalParams& p, long long row, bool ok) {
    WideRowAux x;
    x.lab = -1;
    x.inst_w = 1.f;
#pragma unroll
    for (int j = 0; j < kFuseTargets; ++j) x.tg[j] = 0.f;
    if (ok) {
        const int lik_kind = p.net.lik_kind, k = p.net.k_targets;
        if (lik_kind == NPBNN_LIK_CATEGORICAL && p.labels != nullptr) {
            x.lab = p.labels[row];
            if (p.inst_w) x.inst_w = p.inst_w[row];
        } else if (lik_kind == NPBNN_LIK_GAUSS && p.targets != nullptr) {
#pragma unroll
            for (int j = 0; j < kFuseTargets; ++j)
                if (j < k) x.tg[j] = p.targets[row * k + j];
        }
    }
    return x;
}
__device__ __forceinline__ void wide_row_terms_lean(const EvalParams& p, const float* z, long long row, const float* image, long long classw_off,
                                                    float final_prm, const WideRowAux& aux, WideRowAccLean& A) {
    const NetMeta& net = p.net;
    const int C = net.n_out, lik_kind = net.lik_kind, k = net.k_targets;
    const int act_kind = net.act_kind;
    auto val = [&](int o) -> float {
        const float v = z[o];
        return net.final_act ? act_apply(v, act_kind, final_prm) : v;
    };
    const bool need_softmax = lik_kind == NPBNN_LIK_CATEGORICAL || (p.predict_mode == 2 && net.out_kind == NPBNN_OUT_SOFTMAX);
    float lse = 0.f;
    int best = 0;
    if (need_softmax) {
        float m = -INFINITY;
        for (int o = 0; o < C; ++o) {
            const float v = val(o);
            if (v > m) { m = v; best = o; }
        }
        float se = 0.f;
        for (int o = 0; o < C; ++o) se += __expf(val(o) - m);
        lse = m + log_1_to_n(se);
    }
    if (lik_kind == NPBNN_LIK_CATEGORICAL && p.labels != nullptr) {
        const int lab = aux.lab;
        if (lab >= 0) {
            float wgt = aux.inst_w;
            if (p.use_classw && classw_off >= 0) wgt *= image[classw_off + lab];
            float term = (lab < C ? val(lab) : 0.f) - lse;
            term *= wgt;
            A.ll += (double)term;
            if (p.confusion) atomicAdd(p.confusion + (size_t)lab * C + best, 1u);
        }
    } else if (lik_kind == NPBNN_LIK_GAUSS && p.targets != nullptr) {
#pragma unroll
        for (int j = 0; j < kFuseTargets; ++j)
            if (j < k) {
                const float r = aux.tg[j] - val(j);
                A.s1[j] += (double)r;
                A.s2[j] += (double)r * (double)r;
            }
    }
    if (p.predict_mode && p.y_out != nullptr) {
        float* yo = p.y_out + row * C;
        for (int o = 0; o < C; ++o) {
            float v = val(o);
            if (p.predict_mode == 2) {
                if (net.out_kind == NPBNN_OUT_SOFTMAX) v = __expf(v - lse);
                else if (net.out_kind == NPBNN_OUT_SOFTPLUS_HALF && o >= C / 2) v = softplus_f(v);
            }
            yo[o] = v;
        }
    }
}
template <typename ACC, int KT>
__device__ __forceinline__ void wide_block_partials_t(const EvalParams& p, const ACC& A, double* red, int n_waves, int slot, int n_slots, int cand = 0) {
    const int lik_kind = p.net.lik_kind, k = p.net.k_targets;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lik_kind == NPBNN_LIK_GAUSS) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if (j < k) {
                const double a1 = butterfly_sum_f64(A.s1[j]), a2 = butterfly_sum_f64(A.s2[j]);
                if (lane == 0) { red[wave * kPartialStride + 1 + j] = a1; red[wave * kPartialStride + 1 + NPBNN_MAX_TARGETS + j] = a2; }
            }
        if (lane == 0) red[wave * kPartialStride] = 0.0;
    } else {
        const double s = butterfly_sum_f64(A.ll);
        if (lane == 0) red[wave * kPartialStride] = s;
    }
    __syncthreads();
    const int nvals = lik_kind == NPBNN_LIK_GAUSS ? 1 + 2 * k : 1;
    for (int it = threadIdx.x; it < nvals; it += blockDim.x) {
        const int v = it <= k ? it : 1 + NPBNN_MAX_TARGETS + (it - k - 1);
        double s = 0.0;
        for (int w = 0; w < n_waves; ++w) s += red[w * kPartialStride + v];
        p.partials[((size_t)cand * kPartialStride + v) * n_slots + slot] = s;          // (pass parity 0: [candidate][value][workgroup])
    }
}
// one partial record per workgroup: lanes -> wave (fixed butterfly) -> workgroup (waves in order) -> partials[value][slot of n_slots]
// (every wave of the workgroup calls it; `red`: LDS, n_waves x kPartialStride doubles)
__device__ __forceinline__ void wide_block_partials(const EvalParams& p, const WideRowAcc& A, double* red, int n_waves, int slot, int n_slots) {
    wide_block_partials_t<WideRowAcc, NPBNN_MAX_TARGETS>(p, A, red, n_waves, slot, n_slots);
}

// ------------------------------------------------------------------------------------------------
// the tiled matrix product of one layer
// ------------------------------------------------------------------------------------------------
struct WideGemmArgs {
    const float* A;           // [rows][lda] float32, or the fp16-split copy of X (same bytes per element: per 8 columns, 8 fp16 high
                              // parts then 8 fp16 low parts)
    long long lda;            // floats per row
    int n_row_tiles;          // 16-row tiles of A (A is zero padded to whole tiles)
    int n_units;              // K-units of 32 columns
    int a_half_last;          // 1: A's rows end 16 columns into the last unit - its second piece does not exist and the first is read
                              // in its place (the fragments of those columns are zero)
    int mt_total;             // 16-unit output tiles of the layer
    const float* W;           // the layer's fragment block
    const float* bias;        // [16 * mt_total]
    float* out;               // [rows][ldo]
    long long ldo;
    int act_kind;             // activation applied to the result (hidden layers), -1: none (last layer)
    float act_prm;
    const double* act_prm_dev;   // device chain with trainable slopes: the candidate's slope for this layer, else nullptr
    const PassDesc* pass;     // chain pass: nothing to do when its descriptor says the batch is through (n_cand == 0); else nullptr
    int n_stage;              // stages of the LDS ring (>= 2)
    int a_tiled;              // A is the tile-major fp16-split copy of X (split_x_tiled_kernel): every 1-KiB piece is contiguous
    int k_slices;             // > 1: the contraction is cut into that many slices of K-units, one workgroup each; slice s writes its raw
    int pad_;                 // sums (the bias in slice 0, no activation) to out + s * slice_stride; wide_reduce_kernel adds them up
    long long slice_stride;   // floats
    // fused end (tilings whose waves hold whole rows - one output block, WC = 1 - on one K-slice): behind the product the narrow layers
    // that follow (tail) and the likelihood terms of the rows, in the same launch - no activations through HBM, no further launches
    int fuse;                 // 0 none; 1 tail layers + likelihood (p, image, classw_off; one partial record per row block)
    int n_row_blocks;         // slots of the partial records
    WideTailDesc tail;
    const EvalParams* p;
    const float* image;
    long long classw_off;
    long long cand_stride;    // D > 1 (fused passes of a device chain): candidate j's weight image starts j * cand_stride floats behind W / bias / image
};

#define NPBNN_WVM_(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wide_wait_vm(int younger) {     // wave-uniform: all but the `younger` youngest vector-memory operations are done
    switch (younger) {
        NPBNN_WVM_(0) NPBNN_WVM_(1) NPBNN_WVM_(2) NPBNN_WVM_(3) NPBNN_WVM_(4) NPBNN_WVM_(5) NPBNN_WVM_(6) NPBNN_WVM_(7) NPBNN_WVM_(8) NPBNN_WVM_(9)
        NPBNN_WVM_(10) NPBNN_WVM_(11) NPBNN_WVM_(12) NPBNN_WVM_(13) NPBNN_WVM_(14) NPBNN_WVM_(15) NPBNN_WVM_(16) NPBNN_WVM_(17) NPBNN_WVM_(18)
        NPBNN_WVM_(19) NPBNN_WVM_(20) NPBNN_WVM_(21) NPBNN_WVM_(22) NPBNN_WVM_(23) NPBNN_WVM_(24) NPBNN_WVM_(25) NPBNN_WVM_(26) NPBNN_WVM_(27)
        NPBNN_WVM_(28) NPBNN_WVM_(29) NPBNN_WVM_(30) NPBNN_WVM_(31) NPBNN_WVM_(32) NPBNN_WVM_(33) NPBNN_WVM_(34) NPBNN_WVM_(35) NPBNN_WVM_(36)
        NPBNN_WVM_(37) NPBNN_WVM_(38) NPBNN_WVM_(39) NPBNN_WVM_(40) NPBNN_WVM_(41) NPBNN_WVM_(42) NPBNN_WVM_(43) NPBNN_WVM_(44) NPBNN_WVM_(45)
        NPBNN_WVM_(46) NPBNN_WVM_(47) NPBNN_WVM_(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (more than the counter holds in flight never happens: kWideMaxYounger)
    }
}
#undef NPBNN_WVM_
constexpr int kWideMaxYounger = 48;
constexpr int kWideMaxSlices = 8;    // K-slices a layer's contraction is cut into at most
constexpr int kWideFlush = 8;        // K-units (256 columns) added up by the matrix cores before the sum joins the running total

// RT x CT: 16 x 16 tiles a wave computes (rows x outputs); WR x WC: waves of the workgroup (rows x outputs); F16: fp16-split operands;
// D: weight sets multiplied against the SAME pieces of the rows (the candidates of a speculative chain pass - the resident path's
// idea: narrow networks on many features are bound by the intake of X, so a second candidate's fragments cost a third more bytes and
// no more time per row; fused passes only)
template <int RT, int CT, int WR, int WC, bool F16, int D = 1, int DMAX = D>
__global__ void __launch_bounds__(WR * WC * 64) wide_gemm_kernel(const WideGemmArgs a) {
    constexpr int NW = WR * WC;
    constexpr int XT = WR * RT, WT = WC * CT;            // row tiles / output tiles of the workgroup
    constexpr int PIECES = 2 * (XT + D * WT);            // 1-KiB LDS-DMA pieces per stage
    static_assert(PIECES % NW == 0, "every wave requests the same number of pieces per unit");
    static_assert(D == 1 || WC == 1, "several candidates: fused passes only (every wave holds whole rows)");
    constexpr int PPW = PIECES / NW;
    constexpr int STAGE = (XT + D * WT) * 2048;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.pass != nullptr) {
        const int n_cand = __builtin_amdgcn_readfirstlane(a.pass->n_cand);
        if (n_cand == 0) return;
    }
    // workgroup -> (row block, output block): the output blocks of one row block sit 8 workgroup indices apart - the same XCD (workgroups
    // go round the 8 XCDs), dispatched close together: the second reads the rows' pieces out of the L2 the first filled
    const int n_cb = (a.mt_total + WT - 1) / WT;
    const int n_sl = a.k_slices > 1 ? a.k_slices : 1;
    const int per_grp = 8 * n_cb * n_sl;               // (the slices of a block of the output sit 8 apart as well)
    const int bid = (int)blockIdx.x;
    const int grp = bid / per_grp, rem = bid % per_grp;
    const int cb = (rem / 8) % n_cb, slice = rem / (8 * n_cb), rb = grp * 8 + rem % 8;
    if (rb * XT >= a.n_row_tiles) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int wr = wave / WC, wc = wave % WC;
    const int n_stage = a.n_stage;
    // this workgroup's K-units: [u_begin, u_begin + n_units)
    const int u_begin = (int)((long long)a.n_units * slice / n_sl);
    const int n_units = (int)((long long)a.n_units * (slice + 1) / n_sl) - u_begin;

    // ---- this wave's pieces of a stage: piece p = wave + i * NW; p < 2 XT: rows (tile p >> 1, half p & 1), else fragments ----
    const float* src[PPW];        // this lane's source address for unit 0
    long long ustride[PPW];       // floats per unit (wave-uniform)
    int lds_off[PPW];             // byte offset inside a stage (wave-uniform)
    int half_x[PPW];              // 1: the second piece of a row tile (a_half_last)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + i * NW;
        if (p < 2 * XT) {
            const int t = p >> 1, h = p & 1;
            int T = rb * XT + t;
            if (T > a.n_row_tiles - 1) T = a.n_row_tiles - 1;        // (past the matrix: the last tile again; its results are not stored)
            if (a.a_tiled) {      // tile-major copy: piece (T, u, h) is the 1 KiB at ((T * units + u) * 2 + h) * 256 floats
                src[i] = a.A + (((long long)T * a.n_units) * 2 + h) * 256 + lane * 4;
                ustride[i] = 512;
            } else {
                src[i] = a.A + ((long long)T * 16 + n) * a.lda + 16 * h + 4 * kq;
                ustride[i] = 32;
            }
            lds_off[i] = t * 2048 + h * 1024;
            half_x[i] = h;
        } else {
            const int q = p - 2 * XT;
            const int jc = q / (2 * WT), qq = q % (2 * WT);       // candidate, piece of its fragments
            const int c = qq >> 1, h = qq & 1;
            int mt = cb * WT + c;
            if (mt > a.mt_total - 1) mt = a.mt_total - 1;
            const float* const Wj = a.W + (long long)jc * a.cand_stride;
            if constexpr (F16) {
                src[i] = Wj + ((long long)mt * 2 + h) * 256 + lane * 4;
                ustride[i] = (long long)a.mt_total * 512;
            } else {
                src[i] = Wj + ((long long)h * a.mt_total + mt) * 256 + lane * 4;
                ustride[i] = (long long)a.mt_total * 512;
            }
            lds_off[i] = XT * 2048 + (jc * WT + c) * 2048 + h * 1024;
            half_x[i] = 0;
        }
    }
    int st_in = 0, st_out = 0;                // stage the next request goes to / the next unit is read from
    auto issue = [&](int u) {                 // request unit u into its stage of the ring
        char* const sb = smem + (size_t)st_in * STAGE;
        st_in = st_in + 1 == n_stage ? 0 : st_in + 1;
        const bool last_half = a.a_half_last && u_begin + u == a.n_units - 1;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const float* g = src[i] + (long long)(u_begin + u) * ustride[i];
            if (half_x[i] && last_half) g -= 16;
            dma16(g, sb + lds_off[i]);
        }
    };

    // ---- two levels of float32 accumulation: the matrix cores add into `acc`, which is emptied into `tot` every kWideFlush units
    //      (a long contraction's rounding error grows with the length of ONE chain of additions: thousands of columns would
    //      otherwise cost the last layer's values a digit); `tot` starts from the bias ----
    // DMAX = the most candidates this tiling is built for: ALL its builds accumulate alike, so that a chain's sums do not depend on how many
    // candidates share a pass (and mh_step's single evaluations are those of run_steps' passes, bit for bit)
    constexpr bool TWO = DMAX * RT * CT <= 16 && NW <= 8;   // (the tilings of 32 tiles per wave have no registers for a second set: their long
                                              // contractions are cut into K-slices, whose sums meet in wide_reduce_kernel)
    f32x4 accs[D][RT][CT], tots[TWO ? D : 1][TWO ? RT : 1][TWO ? CT : 1];
    auto& acc = accs[0];            // (the single weight set of the plain forms below)
#pragma unroll
    for (int jc = 0; jc < D; ++jc)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            int mt = cb * WT + wc * CT + ct;
            if (mt > a.mt_total - 1) mt = a.mt_total - 1;
            f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + (long long)jc * a.cand_stride + 16 * mt + 4 * kq);
            if (slice > 0) b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (TWO) { tots[jc][rt][ct] = b; accs[jc][rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                else accs[jc][rt][ct] = b;
            }
        }
    int since_flush = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the bias loads: not counted with the pieces below)

    for (int u = 0; u < n_stage - 1 && u < n_units; ++u) issue(u);
    for (int u = 0; u < n_units; ++u) {
        int ahead = n_units - 1 - u;                      // units requested behind unit u
        if (ahead > n_stage - 2) ahead = n_stage - 2;
        wide_wait_vm(ahead * PPW);
        __syncthreads();                                  // unit u has landed for every wave; every wave is through with unit u - 1
        if (u + n_stage - 1 < n_units) issue(u + n_stage - 1);      // (into the stage unit u - 1 was read from)
        const char* const sb = smem + (size_t)st_out * STAGE;
        st_out = st_out + 1 == n_stage ? 0 : st_out + 1;
        if constexpr (F16) {
            // lane (n, kq) takes feature group kq of the unit: piece kq >> 1, entries 2 (kq & 1) (high parts) and 2 (kq & 1) + 1 (low parts)
            f16x8 xh[RT], xl[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const char* px = sb + (wr * RT + rt) * 2048 + (kq >> 1) * 1024 + ((2 * (kq & 1)) * 16 + n) * 16;
                xh[rt] = *reinterpret_cast<const f16x8*>(px);
                xl[rt] = *reinterpret_cast<const f16x8*>(px + 256);
            }
#pragma unroll
            for (int jc = 0; jc < D; ++jc)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const char* pw = sb + XT * 2048 + (jc * WT + wc * CT + ct) * 2048 + lane * 16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(pw);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(pw + 1024);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accs[jc][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[rt], accs[jc][rt][ct], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accs[jc][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[rt], accs[jc][rt][ct], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) accs[jc][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[rt], accs[jc][rt][ct], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 x[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) x[rt] = *reinterpret_cast<const f32x4*>(sb + (wr * RT + rt) * 2048 + h * 1024 + lane * 16);
#pragma unroll
                for (int jc = 0; jc < D; ++jc)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(sb + XT * 2048 + (jc * WT + wc * CT + ct) * 2048 + h * 1024 + lane * 16);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) accs[jc][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], x[rt][s], accs[jc][rt][ct], 0, 0, 0);
                    }
            }
        }
        if constexpr (TWO) {
            if (++since_flush == kWideFlush) {
                since_flush = 0;
#pragma unroll
                for (int jc = 0; jc < D; ++jc)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) { tots[jc][rt][ct] += accs[jc][rt][ct]; accs[jc][rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
        }
    }

    if constexpr (TWO) {
#pragma unroll
        for (int jc = 0; jc < D; ++jc)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) accs[jc][rt][ct] += tots[jc][rt][ct];
    }
    // ---- activation, store: lane (n, kq) holds the units 4 kq .. 4 kq + 3 of a tile for row n ----
    float prm = a.act_prm;
    if (a.act_prm_dev != nullptr) prm = (float)*a.act_prm_dev;
    if constexpr (WC == 1 && CT <= kTailIn) {
        if (a.fuse) {
            // every wave holds whole rows of this layer's values: the narrow layers behind it from LDS copies of their weights (the ring is
            // free: every wave is through with the last unit after the barrier), the rows' values through a per-wave LDS scratch to one
            // lane per row for the likelihood terms
            const EvalParams& p = *a.p;
            __syncthreads();
            float* const lds = reinterpret_cast<float*>(smem);
            int tail_floats = 0;
            if (a.tail.n_layers > 0) tail_floats = a.tail.lds_bias[a.tail.n_layers - 1] + 16 * a.tail.mt[a.tail.n_layers - 1];
            const int tail_pad = (tail_floats + 3) & ~3;
            for (int jc = 0; jc < D; ++jc) wide_tail_stage(a.tail, lds + jc * tail_pad, tid, NW * 64, a.tail.image + (long long)jc * a.cand_stride);
            __syncthreads();
            const int C = p.net.n_out;
            const int ldz = ((C + 15) & ~15) + 1;            // (odd: the 16 lanes that each read one row of the scratch hit 16 banks)
            float* const zs = lds + D * tail_pad + wave * 16 * ldz;
            double* const red = reinterpret_cast<double*>(zs + (NW - wave) * 16 * ldz + 4);      // (behind every wave's scratch, 8-byte aligned below)
            double* const red8 = reinterpret_cast<double*>((reinterpret_cast<size_t>(red) + 7) & ~(size_t)7);
            const float fprm = p.net.act_prm[p.net.n_layers - 1];
            // ONE copy of the tail and the row terms in the code, run per (candidate, row tile): the tile's accumulators are picked by a
            // switch with constant indices in every case (a run-time index into the register arrays would put them in scratch, and the
            // body unrolled D x RT times is tens of thousands of instructions)
#pragma unroll 1
            for (int jc = 0; jc < D; ++jc) {
                WideRowAccLean A;
                A.ll = 0.0;
#pragma unroll
                for (int j = 0; j < kFuseTargets; ++j) { A.s1[j] = 0.0; A.s2[j] = 0.0; }
#pragma unroll 1
                for (int rt = 0; rt < RT; ++rt) {
                    const int T = rb * XT + wr * RT + rt;
                    if (T >= a.n_row_tiles) continue;
                    const long long row = (long long)T * 16 + lane;
                    const WideRowAux aux = wide_row_aux(p, row, lane < 16 && row < p.n_rows);      // (requested ahead of the tail's arithmetic)
                    f32x4 h[kTailIn];
#pragma unroll
                    for (int ct = 0; ct < kTailIn; ++ct) h[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int item = jc * RT + rt;
#pragma unroll
                    for (int q = 0; q < D * RT; ++q)
                        if (item == q) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) h[ct] = accs[q / RT][q % RT][ct];
                        }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        if (ct < a.mt_total) {
                            if (a.act_kind >= 0) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) h[ct][i] = act_apply(h[ct][i], a.act_kind, prm);
                            }
                        } else {
                            h[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    const int mt_last = wide_tail_layers(a.tail, lds + jc * tail_pad, h, a.mt_total, lane, kq);
#pragma unroll
                    for (int mt = 0; mt < kTailOut; ++mt)
                        if (mt < mt_last) {                                  // (the odd row stride leaves no 16-byte alignment: dword stores)
#pragma unroll
                            for (int i = 0; i < 4; ++i) zs[n * ldz + 16 * mt + 4 * kq + i] = h[mt][i];
                        }
                    if (lane < 16 && row < p.n_rows)
                        wide_row_terms_lean(p, zs + lane * ldz, row, a.image + (long long)jc * a.cand_stride, a.classw_off, fprm, aux, A);
                }
                if (p.partials != nullptr) {
                    __syncthreads();                              // (the sums' scratch: free of the candidate before)
                    wide_block_partials_t<WideRowAccLean, kFuseTargets>(p, A, red8, NW, rb, a.n_row_blocks, jc);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int T = rb * XT + wr * RT + rt;
        if (T >= a.n_row_tiles) continue;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int mt = cb * WT + wc * CT + ct;
            if (mt >= a.mt_total) continue;
            f32x4 v = acc[rt][ct];
            if (a.act_kind >= 0 && n_sl == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i], a.act_kind, prm);
            }
            *reinterpret_cast<f32x4*>(a.out + (long long)slice * a.slice_stride + ((long long)T * 16 + n) * a.ldo + 16 * mt + 4 * kq) = v;
        }
    }
}

typedef void (*wide_gemm_fn_t)(const WideGemmArgs);
struct WideCandState { int prev_t0, prev_n, prev_cnt[kMaxCand]; };      // what the candidate images' last patches covered (wide_cand_*_kernel)

#ifdef NPBNN_KERNELS_WIDE
// the K-slices' sums of a layer, added in slice order (fixed: the same bits every run), then the activation
__global__ void __launch_bounds__(256) wide_reduce_kernel(const float* __restrict__ part, long long slice_stride, int n_slices, float* __restrict__ out,
                                                          long long n_vec4, int act_kind, float act_prm, const double* act_prm_dev, const PassDesc* pass) {
    if (pass != nullptr && pass->n_cand == 0) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec4) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + 4 * i);
    for (int s = 1; s < n_slices; ++s) v += *reinterpret_cast<const f32x4*>(part + (long long)s * slice_stride + 4 * i);
    if (act_kind >= 0) {
        const float prm = act_prm_dev != nullptr ? (float)*act_prm_dev : act_prm;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], act_kind, prm);
    }
    *reinterpret_cast<f32x4*>(out + 4 * i) = v;
}

// The fp16-split copy of X in the order the products read it: per 16-row tile T and K-unit u the two 1-KiB pieces a workgroup's
// LDS-DMA fetches, each contiguous - piece h holds, for lane (row n, entry kq), 16 bytes: entries 0 / 2 the high parts of feature
// groups 2 h / 2 h + 1 of the unit, entries 1 / 3 their low parts (what the 16-rows-x-64-bytes piece of the row-major copy puts into
// LDS).  A tile's whole K-stream is one contiguous run: 1-KiB requests instead of 16 x 64 bytes 4 * F bytes apart (measured,
// tools/microbench_ingest.hip: 1.3-1.5 x the LDS-DMA rate per compute unit).
__global__ void __launch_bounds__(256) split_x_tiled_kernel(const float* __restrict__ X, long long n_rows_pad, int Fp, int n_units,
                                                            const float* __restrict__ x_scale, float* __restrict__ X16w) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;     // one thread per (row, group of 8 features)
    const int groups = n_units * 4;
    if (g >= n_rows_pad * groups) return;
    const long long r = g / groups;
    const int grp = (int)(g % groups), c0 = grp * 8;
    f16x8 hi, lo;
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        const float v = c < Fp ? X[r * Fp + c] * x_scale[c] : 0.f;
        _Float16 h, l;
        split_f16(v, h, l);
        hi[j] = h;
        lo[j] = l;
    }
    const long long T = r >> 4;
    const int n = (int)(r & 15), u = grp >> 2, kg = grp & 3;
    float* piece = X16w + (((T * n_units + u) * 2) + (kg >> 1)) * 256;
    *reinterpret_cast<f16x8*>(piece + ((2 * (kg & 1)) * 16 + n) * 4) = hi;
    *reinterpret_cast<f16x8*>(piece + ((2 * (kg & 1) + 1) * 16 + n) * 4) = lo;
}
#endif  // NPBNN_KERNELS_WIDE

// ------------------------------------------------------------------------------------------------
// the narrow end of a network in one launch (where the product before it could not take it along: several output blocks or K-slices):
// as tiled products of their own each narrow layer would cost a launch, a pass over the activations and a prologue per workgroup for
// a few MFMAs
// ------------------------------------------------------------------------------------------------
struct WideTailArgs {
    const float* A;           // [rows][lda] activations behind the last tiled product (activation applied)
    long long lda;
    int n_row_tiles;
    int kt0;                  // 16-column steps of the first layer's input
    WideTailDesc t;
    float* out;               // [rows][ldo] values of the last layer computed
    long long ldo;
    const PassDesc* pass;
};

#ifdef NPBNN_KERNELS_WIDE
__global__ void __launch_bounds__(512) wide_tail_kernel(const WideTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.pass != nullptr) {
        const int n_cand = __builtin_amdgcn_readfirstlane(a.pass->n_cand);
        if (n_cand == 0) return;
    }
    float* const lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    wide_tail_stage(a.t, lds, tid, blockDim.x);
    __syncthreads();
    for (int T = (int)blockIdx.x * nw + wave; T < a.n_row_tiles; T += (int)gridDim.x * nw) {
        f32x4 h[kTailIn];
        const float* row = a.A + ((long long)T * 16 + n) * a.lda + 4 * kq;
#pragma unroll
        for (int ct = 0; ct < kTailIn; ++ct) h[ct] = ct < a.kt0 ? *reinterpret_cast<const f32x4*>(row + 16 * ct) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int kt = wide_tail_layers(a.t, lds, h, a.kt0, lane, kq);
        float* orow = a.out + ((long long)T * 16 + n) * a.ldo + 4 * kq;
#pragma unroll
        for (int mt = 0; mt < kTailOut; ++mt)
            if (mt < kt) *reinterpret_cast<f32x4*>(orow + 16 * mt) = h[mt];
    }
}
#endif  // NPBNN_KERNELS_WIDE

// ------------------------------------------------------------------------------------------------
// likelihood terms / statistics / predictions from the last layer's values (one thread per data row)
// ------------------------------------------------------------------------------------------------
struct WideLikArgs {
    const EvalParams* p;      // the launch's parameter block (labels, targets, row weights, partial sums, confusion counts, predictions ...)
    const float* z;           // [rows][ldz] last layer's values (no activation applied)
    long long ldz;
    const float* image;       // weight image the values came from (class weights)
    long long classw_off;     // or -1
    const double* final_prm_dev;   // slope of the activation behind the last layer (final_act) of a chain candidate, or nullptr
};

#ifdef NPBNN_KERNELS_WIDE
__global__ void __launch_bounds__(256) wide_lik_kernel(const WideLikArgs a) {
    const EvalParams& p = *a.p;
    if (p.has_pass && p.pass_desc[0].n_cand == 0) return;
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    float fprm = p.net.act_prm[p.net.n_layers - 1];
    if (a.final_prm_dev != nullptr) fprm = (float)*a.final_prm_dev;
    WideRowAcc A;
    A.ll = 0.0;
#pragma unroll
    for (int j = 0; j < NPBNN_MAX_TARGETS; ++j) { A.s1[j] = 0.0; A.s2[j] = 0.0; }
    if (row < p.n_rows) wide_row_terms(p, a.z + row * a.ldz, row, a.image, a.classw_off, fprm, A);
    if (p.partials == nullptr) return;
    __shared__ double red[4 * kPartialStride];
    wide_block_partials(p, A, red, 4, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// candidate image of a device chain with WIDE proposals (more entries than the step's workgroup holds at once: the reference's default
// perturbs 5 % of every layer, np_bnn/BNN_env.py:285 - tens of thousands of entries of a network this path exists for).  The step
// keeps the image itself for narrow proposals (ChainParams::cand_image); for wide ones two launches over all compute units do it
// between the step and the pass: the entries the pass before had patched go back to the committed image's values (which hold the
// accepted proposal, if it was accepted), then the pending proposal's entries are patched in.
// ------------------------------------------------------------------------------------------------
// grid.y = candidate image; the entries every candidate of the pass before had patched go back to the committed image's values in ALL
// of them (an accepted candidate's entries changed the committed image: the others take them over here)
__global__ void __launch_bounds__(256) wide_cand_restore_kernel(const ChainParams* __restrict__ cp, const WideCandState* __restrict__ st, float* __restrict__ cand,
                                                                const float* __restrict__ image, long long cand_stride) {
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    float* const mine = cand + (long long)blockIdx.y * cand_stride;
    const int n_old = st->prev_n;
    for (int j = 0; j < n_old; ++j)
        if (e < st->prev_cnt[j]) restore_image_entry(mine, image, cp->pos[(size_t)(st->prev_t0 + j) * cp->M + e]);
}
// grid.y = candidate j of the pending pass: its proposal's entries patched into its image
__global__ void __launch_bounds__(256) wide_cand_apply_kernel(const ChainParams* __restrict__ cp, WideCandState* __restrict__ st, float* __restrict__ cand,
                                                              long long cand_stride) {
    const ChainParams& c = *cp;
    const PassDesc d = c.pass[0];
    const int j = (int)blockIdx.y;
    const int cnt = j < d.n_cand ? d.cnt[j] : 0;
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (e == 0) {                                           // (read by the NEXT restore launch only)
        if (j == 0) { st->prev_t0 = d.t0; st->prev_n = d.n_cand; }
        st->prev_cnt[j] = cnt;
    }
    if (e >= cnt) return;
    const size_t k = (size_t)(d.t0 + j) * c.M + e;
    patch_image(cand + (long long)j * cand_stride, c.pos[k], c.pscale ? c.pscale[k] : 1.0f, c.pv[(size_t)j * c.M + e], 16);
}
// The same launch when the step leaves the making of the candidates to it (ChainParams::prep_terms): grid.y = candidate j of the pending
// pass, one thread per entry of its proposal - the value (W_cur + delta, reflected at the bounds, masked: spec_entry, the step's own
// arithmetic), its place in the patch list the commit reads, the fp16-range flag, the candidate image's entry, and the entry's prior term
// for the step to add up when it decides the pass (in the order its own preparation adds them: the same log prior to the bit).  The
// step is ONE workgroup: 2.6 k entries x 3 candidates (the default network on 1024 features) cost it 24 us of dependent round trips.
__global__ void __launch_bounds__(256) wide_cand_prepare_kernel(const ChainParams* __restrict__ cp, WideCandState* __restrict__ st, float* __restrict__ cand,
                                                                long long cand_stride) {
    const ChainParams& c = *cp;
    const PassDesc d = c.pass[0];
    const int j = (int)blockIdx.y;
    const int cnt = j < d.n_cand ? d.cnt[j] : 0;
    const int e = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (e == 0) {                                           // (read by the NEXT restore launch only)
        if (j == 0) { st->prev_t0 = d.t0; st->prev_n = d.n_cand; }
        st->prev_cnt[j] = cnt;
    }
    if (e >= cnt) return;
    const size_t k = (size_t)(d.t0 + j) * c.M + e;
    const int i = c.idx[k];
    double term = 0.0;
    if (i >= 0) {
        const double dl = c.delta[k];
        const int pos = c.pos[k];
        const float sc = c.pscale ? c.pscale[k] : 1.0f;
        const double base = c.w_cur[i];
        const double m = c.mask ? c.mask[i] : 1.0;
        const double scale_w = c.prior_scale_w ? c.prior_scale_w[i] : 1.0;
        double his = c.half_inv_s2[0], lsc = c.prior_scale[0];
#pragma unroll
        for (int q = 1; q < kMaxLayers; ++q) {
            const bool past = q < c.net.n_layers && i >= c.net.L[q].w_off;
            his = past ? c.half_inv_s2[q] : his;
            lsc = past ? c.prior_scale[q] : lsc;
        }
        const double v = spec_entry<false>(c.w_bound, c.prior_kind, c.prior_scale_w != nullptr, base, dl, m, scale_w, his, lsc, term);
        c.pv[(size_t)j * c.M + e] = v;
        if (pos < 0 && !(fabs(v * (double)sc) <= (double)kF16Safe)) atomicOr(c.overflow, kFlagF16Range);
        patch_image(cand + (long long)j * cand_stride, pos, sc, v, 16);
    }
    c.prep_terms[(size_t)j * c.M + e] = term;
}
#endif  // NPBNN_KERNELS_WIDE

}  // namespace npbnn
