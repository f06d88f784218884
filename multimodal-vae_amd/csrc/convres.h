// Image-resident conv kernels (convres.hip): tried first by launch_gemm_gather for the conv-shaped problems they are
// compiled for.
#pragma once
#include "gemm.h"

// (mmvae_debug_set("convres", 0) turns the path off)
// 1: launched, 0: this problem is not covered (caller falls back to the generic kernels), < 0: error
int try_launch_convres(const GemmParams& p, hipStream_t stream);

// Image-resident weight gradient (convres_wgrad.hip): both operand tensors of NI images resident in LDS, the whole dW of a
// workgroup in accumulators across a persistent loop over image sets.  1: launched (a slab job was registered in ctx),
// 0: not covered, < 0: error.  (mmvae_debug_set("convres_wgrad", 0) turns the path off)
int try_launch_convres_wgrad(WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx);
