// Weight image packing and the one-off fp16-split copy of the feature matrix
// (part of the device code of the npBNN hot path, see npbnn_kernels.hip.h)
#pragma once
#include "npbnn_common.hip.h"

namespace npbnn {

// ------------------------------------------------------------------------------------------------
// pack: float64 packed weights -> fragment image
//   frag_l[((kt*MT + mt)*64 + lane)*4 + s] = W_l[o = 16mt + (lane&15)][c = 16kt + 4(lane>>4) + s]
//   (bias column excluded, zero outside the matrix); bias_l[o] = W_l[o][0] when the layer has a bias.
//   Layer 0 with a column override (data_transform_obj, BNN_env.py:14-17): an overridden feature column
//   is the constant v_c for every row, so its contribution v_c*W0[o][c] moves into the bias and the
//   fragment entry becomes 0 - no extra pass over X.
// ------------------------------------------------------------------------------------------------
// fp16 split of a float: hi = fp16(v), lo = fp16(v - hi); hi + lo carries ~22 significant bits of v
// The float is pinned in a register first: callers hand in (float)(double expression), and a compiler that folds the two
// narrowing conversions into one double -> fp16 rounding picks the other neighbour when the float lands exactly between two
// fp16 values (about one entry in 8000) - the same weight would then be split differently by different kernels.
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    asm volatile("" : "+v"(v));
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

constexpr float kF16Safe = 60000.0f;   // |value| above this does not survive fp16 (max 65504)
// bits of the flag word the packing / chain kernels raise (EvalParams / ChainParams `overflow`)
constexpr int kFlagF16Range = 1;       // a scaled layer-0 weight left the fp16 range: the caller repeats on the float32 path
constexpr int kFlagStructure = 2;      // a weight is not zero where the layer-0 block structure says there are none
constexpr int kFlagBadIndex = 4;       // a pre-drawn weight index of a chain batch lies outside the network

// 16-byte fragment entries of a layer in the image (layer 1 on fp16-split products: a high and a low block per K-step of two
// layer-0 tiles; its output is one tile)
__host__ __device__ inline int layer_frag_items(const NetMeta& net, int l) {
    if (l == 1 && net.l1_f16) return ((net.L[0].mt + 1) / 2) * 2 * 64;
    return net.L[l].kt * net.L[l].mt * 64;      // (layer 0 with fewer than 16 rows per tile in the image: the items of the padding rows write nothing)
}
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
        const int n_pieces = layer_frag_items(net, l);
        if (piece < n_pieces) {
            const int lane = piece & 63;
            const int tile = piece >> 6;
            const int ld = L.in_dim + L.has_bias;
            if (l == 1 && net.l1_f16) {        // (NetMeta::l1_f16) entry ((q*2 + part)*64 + lane): part of W_1[unit at row lane&15][the 8 inputs of (q, lane>>4)]
                const int part = tile & 1, q = tile >> 1, kq = lane >> 4;
                const int o = L.out_perm ? tile_pos(lane & 15) : (lane & 15);
                f16x8 v;
                for (int e = 0; e < 8; ++e) {
                    const int cp = 16 * (2 * q + (e >> 2)) + 4 * kq + (e & 3);     // layer-0 output position 8 kq + e of step q ...
                    const int c = net.l0_rows < 16 ? l0_unit_at(cp, net.l0_rows, L.in_dim) : cp;      // ... and the unit that sits there
                    float wv = 0.f;
                    if (o < L.out_dim && c >= 0 && c < L.in_dim) wv = (float)w[L.w_off + (long long)o * ld + L.has_bias + c];
                    if (overflow && !(fabsf(wv) <= kF16Safe)) atomicOr(overflow, kFlagF16Range);
                    _Float16 hi, lo;
                    split_f16(wv, hi, lo);
                    v[e] = part ? lo : hi;
                }
                *reinterpret_cast<f16x8*>(image + L.frag_off + (long long)piece * 4) = v;
                return;
            }
            if (l == 0 && net.l0_f16) {
                const int part = tile & 1, rest = tile >> 1;
                const int mt = rest % L.mt, ks = rest / L.mt;
                const int rows = net.l0_rows, u = lane & 15;
                const int o = u < rows ? rows * mt + u : 0x40000000;                      // (rows < 16: see NetMeta::l0_rows)
                const int c0 = 32 * ks + 8 * (lane >> 4);
                const bool stored = ks >= net.l0_begin[mt] && ks < net.l0_end[mt] && u < rows;       // (block structure: see NetMeta)
                f16x8 v;
                for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
                if (o < L.out_dim) {
                    const double* row = w + L.w_off + (long long)o * ld + L.has_bias;
                    for (int j = 0; j < 8; ++j) {
                        const int c = c0 + j;
                        if (c < L.in_dim) {
                            const bool overridden = (col_override != nullptr && !isnan(col_override[c]));
                            const float wv = overridden ? 0.f : (float)(row[c] * (double)w_scale[c]);
                            if (overflow && !(fabsf(wv) <= kF16Safe)) atomicOr(overflow, kFlagF16Range);
                            if (overflow && !stored && row[c] != 0.0) atomicOr(overflow, kFlagStructure);
                            _Float16 hi, lo;
                            split_f16(wv, hi, lo);
                            v[j] = part ? lo : hi;
                        }
                    }
                }
                if (stored)      // (slot, part, feature group, row): 4 * rows entries per part - 64, entry = lane, with all 16 rows
                    *reinterpret_cast<f16x8*>(image + L.frag_off +
                                              ((long long)((net.l0_base[mt] + ks - net.l0_begin[mt]) * 2 + part) * (4 * rows) + (lane >> 4) * rows + u) * 4) = v;
                return;
            }
            const int mt = tile % L.mt, kt = tile / L.mt;
            const int o = L.out_perm ? tile_pos(lane & 15) : 16 * mt + (lane & 15);        // (unit at this output position)
            const int c0 = 16 * kt + 4 * (lane >> 4);
            const bool in_perm = l > 0 && net.L[l - 1].out_perm;
            const bool stored = l > 0 || (kt >= net.l0_begin[mt] && kt < net.l0_end[mt]);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (o < L.out_dim) {
                const double* row = w + L.w_off + (long long)o * ld + L.has_bias;
                for (int s = 0; s < 4; ++s) {
                    int c = in_perm ? tile_pos(c0 + s) : c0 + s;                               // (input unit at this position)
                    if (l == 1 && net.l0_f16 && net.l0_rows < 16) c = l0_unit_at(c0 + s, net.l0_rows, L.in_dim);
                    if (c >= 0 && c < L.in_dim) {
                        bool overridden = (l == 0 && col_override != nullptr && !isnan(col_override[c]));
                        v[s] = overridden ? 0.f : (float)row[c];
                        if (overflow && !stored && row[c] != 0.0) atomicOr(overflow, kFlagStructure);
                    }
                }
            }
            if (l > 0) *reinterpret_cast<f32x4*>(image + L.frag_off + (long long)piece * 4) = v;
            else if (stored) *reinterpret_cast<f32x4*>(image + L.frag_off + ((long long)(net.l0_base[mt] + kt - net.l0_begin[mt]) * 64 + lane) * 4) = v;
            return;
        }
        piece -= n_pieces;
    }
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerMeta& L = net.L[l];
        const int nb = 16 * L.mt;
        if (piece < nb) {
            int o = L.out_perm ? tile_pos(piece) : piece;
            if (l == 0 && net.l0_f16 && net.l0_rows < 16) o = l0_unit_at(piece, net.l0_rows, L.out_dim);
            double b = (net.pad_masked && l == net.n_layers - 1) ? (double)kPadLogit : 0.0;
            if (o >= 0 && o < L.out_dim) {
                b = 0.0;
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
            image[L.bias_off + piece] = (float)b;
            return;
        }
        piece -= nb;
    }
    if (with_classw && net.classw_off >= 0 && piece < kResidentMaxWidth) {
        image[net.classw_off + piece] = (class_w != nullptr && piece < net.n_out) ? (float)class_w[piece] : 1.0f;
    }
}

__host__ __device__ inline int pack_item_count(const NetMeta& net, bool with_classw) {
    int total = with_classw ? kResidentMaxWidth : 0;
    for (int l = 0; l < net.n_layers; ++l) total += layer_frag_items(net, l) + 16 * net.L[l].mt;
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
                                                        float* __restrict__ w_scale, const int* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Fp) return;
    const float m = __uint_as_float(absmax[c]);
    int e = 0;
    if (m > 0.f && isfinite(m)) (void)frexpf(m, &e);
    if (shift) e -= shift[c];           // (a heavy-tailed column: its largest entry lands below 2^shift instead of below 1)
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

// How well does the fp16 pair represent column c?  Inside fp16's normal range the pair keeps 22 bits of an entry - a RELATIVE error
// like float32's own (two bits short of it) and not counted here (kF16PairRel).  Below it, an entry's absolute error |x' - (hi + lo)|
// is bounded by half the fp16 subnormal spacing, 2^-25 in scaled units - fine while the column's typical entry is within a few powers
// of two of its largest, but with the column's largest entry scaled to just under 1 a heavy-tailed column (one 1e4 outlier over values
// of 1e-2, log-normal features, ...) puts its typical entries where that error is per cent of the value.  Per column the kernel
// returns the largest counted error of any entry and two yardsticks: the mean |x'| (what the column contributes with: sum of |x'|
// clamped at 2^12, as a 2^-28 fixed-point integer) and the TYPICAL |x'| - the geometric mean of the non-zero entries (zeros are
// exact), from the sum of log2 |x'| as a 2^-16 fixed-point integer; integers, so that the sums do not depend on the order of the
// atomics.  The host (column_quality) asks for error <= 2^-17 of the mean (kF16QualityTol, the bound of rounds 3-4) AND <= 2^-12 of
// the typical entry (kF16TypicalTol: a mean carried by a few outliers must not hide that every other entry is down to a few bits).
// A column past either bound gets its scale MOVED UP by a power of two (ensure_scales: the largest entry stays below
// 2^kF16MaxShift - fp16 has 15 powers of two above 1 that the max-scaled copy leaves unused; the weights' scale moves down by the
// same factor, products unchanged) and is measured again; the cap keeps what a weight's own 2^-25 floor turns into on the row of the
// largest entry at 2^-13.
constexpr float kF16QualityTol = 7.62939453125e-06f;        // 2^-17 of the column's mean |value| (normal data: 2^-21.4; log-normal sigma 1: 2^-18.7;
                                                            // Student t3: 2^-18.1; before any move: log-normal sigma 3: 2^-11.7, one 1e4 outlier
                                                            // over N(0, 1e-2): 2^-10)
constexpr float kF16TypicalTol = 2.44140625e-04f;           // 2^-12 of the column's typical |value|
constexpr float kF16PairRel = 4.76837158203125e-07f;        // 2^-21: an error within this x |entry| is the pair's own 22-bit rounding
constexpr int kF16MaxShift = 12;                            // scaled entries stay below 2^12 (fp16: 65504)
#ifdef NPBNN_KERNELS_MAIN
__global__ void __launch_bounds__(256) split_quality_kernel(const float* __restrict__ X, long long n_rows, int Fp, const float* __restrict__ x_scale,
                                                            unsigned* __restrict__ max_err, unsigned long long* __restrict__ sum_abs,
                                                            unsigned long long* __restrict__ sum_log2, unsigned long long* __restrict__ n_nonzero) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Fp) return;
    const long long r0 = (long long)blockIdx.y * 1024;
    long long r1 = r0 + 1024;
    if (r1 > n_rows) r1 = n_rows;
    const float sc = x_scale[c];
    float worst = 0.f;
    unsigned long long sum = 0, cnt = 0;
    long long lsum = 0;
    for (long long r = r0; r < r1; ++r) {
        const float v = X[r * Fp + c] * sc;
        _Float16 h, l;
        split_f16(v, h, l);
        const float e = fabsf(v - ((float)h + (float)l));      // (hi + lo spans at most 23 bits: the sum is exact in float32)
        const float a = fabsf(v);
        worst = (e > worst && e > kF16PairRel * a) ? e : worst;
        sum += (unsigned long long)((double)(a < 4096.f ? a : 4096.f) * 268435456.0);
        if (a > 0.f && a < INFINITY) {
            lsum += (long long)(log2f(a) * 65536.0f);
            ++cnt;
        }
    }
    atomicMax(max_err + c, __float_as_uint(worst));
    atomicAdd(sum_abs + c, sum);
    atomicAdd(sum_log2 + c, (unsigned long long)lsum);         // (two's complement: the signed sum, whatever the order)
    atomicAdd(n_nonzero + c, cnt);
}
#endif  // NPBNN_KERNELS_MAIN

}  // namespace npbnn
