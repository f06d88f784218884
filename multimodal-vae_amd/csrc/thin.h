// Direct kernels for the thin last ConvTranspose2d(32 -> 1 or 3, k4, s2, p1) of the image decoders. See thin.hip.
#pragma once
#include "common.h"

struct ConvTLastFwdArgs {
    const bf16* act;         // activated input, NHWC [G*B][IH][IW][Cin]
    const float* w;          // fp32 weight (Cin, Cout, 4, 4) straight from the parameter buffer
    int G, B, IH, IW, Cin, Cout;
    const float* target;     // NCHW [B][Cout][2IH][2IW] shared by all groups, or null
    float* logits;           // NCHW [G*B][Cout][2IH][2IW] or null
    float* recon;            // sigmoid(logits) or null
    float* dlogit;           // coef[g] * dBCE/dlogit or null
    float coef[4];
    float* loss_sum;         // [MMVAE_LOSS_SLOTS][16]: slot row, column g += BCE sums, or null
};
int launch_convt_last_fwd(const ConvTLastFwdArgs& a, hipStream_t s);

struct ConvTLastDgradArgs {
    const float* dlogit;     // NCHW [G*B][Cout][2IH][2IW]
    const float* w;
    int G, B, IH, IW, Cin, Cout;
    const bf16* r;           // raw (pre-BN) input of the layer, NHWC [G*B][IH][IW][Cin]
    const float2* affine;    // [G][Cin]
    const float2* meanrstd;  // [G][Cin]
    int act;
    bf16* db;                // out: grad wrt the BN output after the d-activation, same layout as r
    float2* red;             // [G][MMVAE_STAT_SLOTS][Cin] += (sum db, sum db*xhat)
};
int launch_convt_last_dgrad(const ConvTLastDgradArgs& a, hipStream_t s);
