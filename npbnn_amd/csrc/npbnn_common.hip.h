// Shared definitions: layout constants, parameter blocks, lane-level helpers
// (part of the device code of the npBNN hot path, see npbnn_kernels.hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "npbnn_hip.h"

namespace npbnn {

constexpr int kMaxLayers = NPBNN_MAX_LAYERS;
// The LDS-resident path (eval_kernel) holds layers of up to kResidentMaxWidth nodes; wider ones - up to NPBNN_MAX_WIDTH - run on the
// weight-streamed path (npbnn_wide.hip.h), as does any network whose image does not fit a compute unit's LDS.
constexpr int kResidentMaxWidth = 128;
constexpr int kMaxMT = kResidentMaxWidth / 16;   // 16-unit tiles per layer of the resident path
#ifndef NPBNN_RING
#define NPBNN_RING 4
#endif
constexpr int kRing = NPBNN_RING;              // X ring slots (1 KiB each) per wave; kRing-1 pieces stay in flight
constexpr int kMaxWavesPerBlock = 16;
// Per-wave row-aux slots (labels / instance weights / targets of a 16-row tile, fetched ahead of the tile's X pieces).  Their
// number and size depend on the data set and the network: the host lays them out (WaveLayout) and hands the numbers to the
// kernel, so that nothing is reserved that a run does not use - at config 2 this is what lets a 13th wave fit the LDS.
struct WaveLayout {
    int aux_slots;   // 2 when the ring never holds pieces of more than the next tile, else 4 (a power of two)
    int aux_sz;      // bytes per slot
    int off_w;       // instance weights inside a slot (labels, when present, sit at 0)
    int off_t;       // targets inside a slot
    int wave_lds;    // bytes of LDS per wave: ring + slots (+ row scratch of the float64 row-wise likelihoods)
};
// likelihoods that combine several outputs of one row (predicted sigma, count data) exchange them through 1 KiB of LDS
__host__ __device__ inline bool lik_needs_row_scratch(int lik_kind) {
    return lik_kind >= NPBNN_LIK_GAUSS_PRED_SIGMA && lik_kind <= NPBNN_LIK_NEGBIN_BASE10;
}
__host__ __device__ inline WaveLayout make_wave_layout(bool labels, bool inst_w, int k_targets, int kt0, int lik_kind) {
    WaveLayout L;
    L.off_w = labels ? 64 : 0;
    L.off_t = L.off_w + (inst_w ? 64 : 0);
    L.aux_sz = L.off_t + 64 * k_targets;
    L.aux_slots = kt0 >= kRing ? 2 : 4;
    L.wave_lds = kRing * 1024 + L.aux_slots * L.aux_sz + (lik_needs_row_scratch(lik_kind) ? 1024 : 0);
    return L;
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
    // Narrow hidden layers (one output tile, not the last layer): unit u sits at position 4 (u % 4) + u / 4 of the tile instead of u
    // (tile_pos).  A lane (row n, k-group kq) holds positions 4 kq + i in register i, and the next layer's i-th float32 MFMA contracts
    // the positions {i, 4 + i, 8 + i, 12 + i}: with the units transposed like this the first ceil(out / 4) registers hold all of them,
    // and the next layer skips the MFMAs of the others - they would multiply the zeros of the padding (in_live of that layer).
    int out_perm;  // this layer's units are placed like that
    int in_live;   // float32 MFMAs per input tile this layer needs (4 unless the layer before it has out_perm)
};
__host__ __device__ inline int tile_pos(int u) { return 4 * (u & 3) + (u >> 2); }      // (its own inverse: a 4 x 4 transpose)

struct NetMeta {
    int n_layers;
    int image_floats;   // total floats of the image (multiple of 256)
    int classw_off;     // float offset of class weights (kResidentMaxWidth floats; only present when class weights are set, else -1)
    int act_kind, out_kind, lik_kind, n_out, k_targets;
    int final_act;      // apply the activation to the last layer's output too (RunHiddenLayer on its own)
    int l0_f16;         // layer 0 runs on the fp16-split path (x = xh + xl, w = wh + wl; 3 f16 MFMAs, f32 accumulate)
    int pad_masked;     // the padding outputs of the last layer (n_out .. 16 * mt - 1) carry a bias of kPadLogit: they drop out of
                        // the softmax by themselves (exp -> 0, never the maximum) and the epilogue needs no per-output predicate
    // Block structure of layer 0 (masks of create_mask, np_bnn/BNN_lib.py:16-47: consecutive input columns wired to one block of
    // nodes).  K-unit = what one step of the layer-0 loop contracts: 32 features on the fp16-split path, 16 on the float32 path.
    // Output tile mt (16 nodes) has weights only in the K-units l0_begin[mt] .. l0_end[mt]-1; its fragments sit in the image one
    // after the other, the first of them `l0_base[mt]` fragment slots into the layer-0 block (dense layer: begin 0, end = all
    // units, base = mt * units).  The loop skips (unit, tile) pairs outside the range: no fragment reads, no MFMAs - and the
    // skipped weights, being zero, would have added nothing, so the sums are the dense ones bit for bit.
    int l0_begin[kMaxMT], l0_end[kMaxMT], l0_base[kMaxMT];
    LayerMeta L[kMaxLayers];
    float act_prm[kMaxLayers];
    int l1_f16;         // layer 1 runs on fp16-split products too (tanh networks with at least two layer-0 tiles and narrow later layers, on
                        // the fp16-split image): its inputs are tanh values, |h| <= 1, split h = hh + hl like the data, its weights
                        // like layer 0's; K-step q contracts the layer-0 tiles 2q and 2q+1 - lane (n, kq) holds units 16 t + 4 kq + 0..3
                        // of each, which are positions 8 kq + 0..7 of the step.  3 MFMAs of 16 cycles per step instead of 8 of 32.
    int slope_off;      // >= 0: float offset of kMaxLayers slots for per-candidate activation slopes (NPBNN_OPT_TRAINABLE_SLOPES; a chain
                        // pass writes each candidate's slopes into its LDS image copy), else -1
    int l0_rows;        // output units per layer-0 tile IN THE IMAGE of the fp16-split path: 16, or - a dense first layer of three or more
                        // tiles whose width is not a multiple of 16 - ceil(width / tiles): unit o then sits at position 16 (o / rows) + o % rows
                        // instead of o, a fragment slot holds rows instead of 16 rows (2 parts x 4 feature groups x rows x 16 bytes), and the
                        // lanes of the padding rows read the tile's last real row (finite values that meet zero weights in layer 1).
                        // The reference's default [50, 5] on 256 features: 57 KB instead of 70 KB per candidate - two fit the LDS
};
// the layer-0 unit at output position `pos` of such an image (-1: a padding position), and the other way round
__host__ __device__ inline int l0_unit_at(int pos, int rows, int out_dim) {
    const int t = pos >> 4, u = pos & 15;
    if (u >= rows) return -1;
    const int o = t * rows + u;
    return o < out_dim ? o : -1;
}
__host__ __device__ inline int l0_pos_of_unit(int o, int rows) { return (o / rows) * 16 + o % rows; }
constexpr int kPosCompact = 0x40000000;      // (image positions of fp16-split entries, bit 30: the low part sits 32 * l0_rows halves behind the high part, not 512)

constexpr float kPadLogit = -3.0e38f;     // finite: (pad - max) stays finite, exp of it is exactly 0
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

// Group pass (npbnn_chains_run_batched): the candidates of a launch belong to DIFFERENT chains - candidate j is the current proposal
// of chain j: that chain's weight image patched with that chain's pre-drawn perturbation, its sums go to that chain's step.  The
// chains are replicas of one model (MC3.__init__, np_bnn/BNN_mc3.py:55-75) over the same resident X: one streaming read of the
// matrix serves all of them, and every candidate slot is useful whatever the acceptance rate.
struct GroupSlot {
    const ChainParams* chain;     // the chain's step parameters (its step runs in one workgroup of the launch)
    const PassDesc* pass;         // [2] its pass descriptors (inside its own EvalParams block), by pass parity
    const float* image;           // its committed weight image
    const double* pv;             // [2][kMaxCand][M] its candidate patch values (slot 0 of each parity is used)
    const int* pos;               // [K][M] image positions of its pre-drawn entries
    const float* pscale;          // [K][M] fp16-split scales of them, or nullptr
    double* partials;             // [2][kMaxCand][kPartialStride][evaluating workgroups]
    int M, pad_;
};

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
    WaveLayout lay;           // per-wave LDS layout of this launch (host-computed)
    int pad_lay_;
    const ChainParams* chain;  // overlapped chain schedule: the last workgroup of the launch runs chain_step (else nullptr)
    int sync_mode;             // overlapped schedule with the launches alternating between two streams: no kernel boundary orders a
    int share_rot;             // launch after the one before it - device-side flags do (ChainDev.prepared / .done, npbnn_chain.hip.h)
                               // share_rot (persistent overlapped launch): n_tiles % evaluating workgroups - in pass L workgroup b takes the
                               // tiles (and the partial-sum slot) of share (b + L * share_rot) % workgroups, so that the shares with one tile
                               // more - a second round for one wave, which then runs alone - go round instead of holding up the same
                               // workgroups in every pass (the sums are the same, slot for slot)
                               // 1: the step of launch L decides pass L - 1 while pass L is evaluated (NPBNN_SCHED_OVERLAP2 / _PERSIST);
                               // 3: the same with the next pass prepared ahead for every outcome (spec_round; NPBNN_SCHED_PERSIST_SERIAL):
                               //    the descriptor names the patch values the pass reads (pad[2]) and, after an accept, the accepted entries (pad[1])
    unsigned long long* stamps;   // diagnostics only (NPBNN_EVAL_STAMPS=1 in npbnn_time_pass): [workgroup][8] wall-clock stamps, else nullptr
    const double* cand_slopes;    // chain pass with trainable activation slopes: [2][kMaxCand][kMaxLayers] slopes of the candidates of the
                                  // two passes in the pipeline (SlopeState::cand, written by the step), else nullptr
    int group_n;                  // > 0: group pass of that many chains (= the candidates of the build); the last group_n workgroups
    int pad_group_;               // of the launch run the chains' steps
    GroupSlot group[kMaxCand];
    NetMeta net;
};

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// max of two numbers in ONE instruction: fmaxf in a kernel compiled in IEEE mode first canonicalises both operands (v_max x, x, x) so
// that a signalling NaN comes out quiet - two more vector instructions per max, for inputs that are never NaN here.  The median of
// (a, b, +inf) is that max.  (Not inline assembly: the compiler's hazard recogniser does not look inside it, and a hand-written
// v_max right behind the MFMA that produces its operand read the register too early - chains that differed from run to run.)
__device__ __forceinline__ float max_nn(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }

__device__ __forceinline__ float act_apply(float z, int kind, float prm) {
#ifdef NPBNN_EXP_NO_ACT      // timing experiment only: activation = identity
    return z;
#endif
    switch (kind) {
        case NPBNN_ACT_RELU: return max_nn(z, 0.f);                                 // BNN_lib.py:51
        case NPBNN_ACT_LEAKY: return z < 0.f ? prm * z : z;                         // BNN_lib.py:55
        case NPBNN_ACT_SWISH: return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));   // BNN_lib.py:60
        // BNN_lib.py:65 (exp-form tanh).  __expf(2z) is exp2(2z * log2(e)); written as exp2(z * 2 log2(e)) it is one multiply less and
        // the same bits (doubling is exact: both products are the same real number, rounded once)
        default: return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z * 2.8853900817779268f) + 1.f);
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
// (after a lane swap the compiler no longer knows its operands to be canonical numbers and turns max_nn back into a canonicalising max -
// three instructions; the median with the largest FINITE float stays one v_med3 and is the same maximum for every finite pair)
__device__ __forceinline__ float max_fin(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.402823466e38f); }
__device__ __forceinline__ float quad_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = max_fin(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max_fin(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// natural logarithm of a number in [1, 2^20] (the softmax denominator after the maximum has been taken out: between 1 and the number
// of outputs): the hardware's log2 (1 ulp) times ln 2 - two instructions; logf's general form spends ten more on denormal inputs and a
// correction term that such arguments never need
__device__ __forceinline__ float log_1_to_n(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
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

// v += v[lane ^ 32]; ^ 16; ^ 8; ^ 4; ^ 2; ^ 1 - the xor butterfly the step's float64 reductions have always used (every lane ends with the
// total; the partners and their order are part of the result's bits), without its twelve ds_bpermute round trips per sum: the halves
// and rows meet through the permlane swaps, lanes inside a row through DPP (xor 8 is a rotation by 8; xor 4 is the rotation by 4 one way
// or the other, chosen by the lane's bit 2; xor 2 and xor 1 are quad permutations).  Same partners, same additions, same bits.
__device__ __forceinline__ double butterfly_sum_f64(double v) {
    {
        const u32x2 rl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const u32x2 rh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    {
        const u32x2 rl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const u32x2 rh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
    }
    v += dpp_f64<0x128>(v);                                   // row_ror:8
    {
        const double down = dpp_f64<0x124>(v);                // row_ror:4: the value of lane - 4 (mod 16)
        const double up = dpp_f64<0x12c>(v);                  // row_ror:12: the value of lane + 4 (mod 16)
        v += (threadIdx.x & 4) ? down : up;
    }
    v += dpp_f64<0x4e>(v);                                    // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0xb1>(v);                                    // quad_perm [1, 0, 3, 2]
    return v;
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
// the same with agent-scope coherence (sc1): the lines come from the memory side whatever this XCD's L2 and this CU's L1 hold
__device__ __forceinline__ void dma16_coherent(const float* g, char* l) {
    __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 16, 0, 16);
}
__device__ __forceinline__ void dma4(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 4, 0, 0);
}

}  // namespace npbnn
