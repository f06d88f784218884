// Ring-staged weight-gradient kernel (wgrad_ring.hip): launcher hooks used by gemm.hip
#pragma once
#include "gemm.h"

// 0: no kernel is compiled for this problem (the caller falls back to the streamed kernel); 1: launched; < 0: error.
// The partial copies are owed to launch_wgrad_ring_reduce (through ctx->ring_jobs)
int try_launch_wgrad_ring(const WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx);
// sums the partial copies of the ring kernel's launches into the packed gradients, one launch (same `only_own` rule as
// launch_wgrad_reduce)
int launch_wgrad_ring_reduce(WgradSlabCtx* ctx, hipStream_t stream, bool only_own);
