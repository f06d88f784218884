// Image-resident conv kernels (convres.hip): tried first by launch_gemm_gather for the conv-shaped problems they are
// compiled for.
#pragma once
#include "gemm.h"

// (mmvae_debug_set("convres", 0) turns the path off)
// 1: launched, 0: this problem is not covered (caller falls back to the generic kernels), < 0: error
int try_launch_convres(const GemmParams& p, hipStream_t stream);
