// Generic strided fp32 GEMM on the fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).
// Used where the model family keeps the reference's own arithmetic: the MNIST MLPs (mnist_f32.hip) and the COCO caption
// GRUs (coco_text.hip), whose 102-step recurrences would amplify bf16 operand rounding.
#pragma once
#include "common.h"

// C[m][n] (+)= sum_k A(m,k) B(k,n) (+ bias[n]) (+ addm[m][n])
struct F32Gemm {
    const float* A; long long a_rs, a_cs;      // A(m,k) = A[m*a_rs + k*a_cs]   (a_rs = 0 broadcasts one row)
    const float* B; long long b_rs, b_cs;      // B(k,n) = B[k*b_rs + n*b_cs]
    int M, N, K;
    float* C; long long ldc;
    const float* bias;                         // [N] or null
    const float* addm; long long ldadd;        // [M][ldadd] matrix added to the product, or null
    int accumulate;                            // 1: C += result (single writer per element unless ksplit > 1)
    int ksplit;                                // > 1: K is cut into `ksplit` chunks over blockIdx.z, the chunks are added to
                                               // C with float atomics (C must hold the value to accumulate onto, e.g. 0)
};
int gemm_f32(const F32Gemm& g, hipStream_t s);
// two independent GEMMs (no grid-level K split) in ONE launch
int gemm_f32_pair(const F32Gemm& a, const F32Gemm& b, hipStream_t s);
