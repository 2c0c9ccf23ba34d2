// Device code of the npBNN hot path for gfx950 (MI355X, CDNA4).
//
//   npbnn_common.hip.h  layout constants, parameter blocks (NetMeta, EvalParams, PassDesc), lane-level helpers
//   npbnn_pack.hip.h    pack_weights_kernel: float64 packed weights -> MFMA "fragment image" (float32, or fp16 hi/lo pairs for
//                       layer 0) + padded biases; col_absmax / col_scale / split_x kernels: one-off fp16-split copy of X
//   npbnn_chain.hip.h   finalize_kernel (fixed-order float64 reduction of the partial sums -> log-likelihood, sigma, moments);
//                       chain_step: decide the candidates of a pass in iteration order, commit, prepare the next pass
//   npbnn_wide.hip.h    the weight-streamed path for networks that do not fit the LDS: wide_gemm_kernel (a layer as a tiled matrix
//                       product, both operands through an LDS ring), wide_lik_kernel, wide_pack_kernel, wide_cand_kernel
//   npbnn_eval.hip.h    eval_kernel<MT0, MTI, F16, D, LK>: fused forward pass of the whole MLP + likelihood terms for D weight
//                       sets against one streaming read of X (+ confusion counts, + prediction output); in the overlapped chain
//                       schedule its last workgroup runs chain_step
//
// Mapping to the reference (np_bnn 0.1.23): the layer loop of MCMC.mh_step (BNN_env.py:449-473) /
// RunPredict (BNN_lib.py:245-256): RunHiddenLayer -> MatrixMultiplicationD (+ bias = column 0) -> ActFun.eval
// (BNN_lib.py:184-193,154-162,83-87); output functions (BNN_lib.py:166-182); likelihoods
// (BNN_lib.py:100-143, BNN_lik.py:5-66); accuracy reductions (BNN_lib.py:195-233); accept test and proposals
// (BNN_env.py:493-494, BNN_mcmc.py:57-69).
//
// Design:
//  * a wavefront owns a 16-row tile of X and computes the TRANSPOSED problem  H^T = W . X^T  with 16x16 MFMA tiles
//    (v_mfma_f32_16x16x32_f16 on fp16 hi/lo pairs for layer 0, v_mfma_f32_16x16x4_f32 elsewhere): A = weight fragment,
//    B = X^T, so the accumulator of layer l (unit on the register/lane-group index, data row on lane&15) is directly the
//    B operand of layer l+1 - the whole MLP chains through the matrix cores with no data movement between layers; bias
//    is the initial accumulator; activations are elementwise on accumulators.
//  * X tiles arrive by LDS-DMA (global_load_lds_dwordx4, one 1-KiB piece = 16 rows x 64 bytes per wave-instruction) into
//    a private per-wave ring; the wave that issued a piece waits for it with a counted s_waitcnt vmcnt(N); no workgroup
//    barrier in the main loop.
//  * all weights live in LDS as lane-linear fragment images (ds_read_b128, conflict free), one per candidate, staged once
//    per persistent workgroup.
//  * softmax / log-likelihood: lane reductions by permlane swaps and DPP, per-row terms in float32, every cross-row sum
//    in float64, one partial per workgroup and candidate, fixed order -> deterministic.
#pragma once
#include "npbnn_common.hip.h"
#include "npbnn_pack.hip.h"
#include "npbnn_chain.hip.h"
#include "npbnn_eval.hip.h"
#include "npbnn_wide.hip.h"
