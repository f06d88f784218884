// Direct kernels for the thin last ConvTranspose2d(32 -> 1 or 3, k4, s2, p1) of the image decoders. See thin.hip.
#pragma once
#include "common.h"
#include "elementwise.h"

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

// Fused tail of the image decoder for Cout = 1 (multimnist/model.py:206-216 + multimnist/train.py:75-76 and their backward):
//   BatchNorm finalize of the producer layer (tables + running statistics) -> BN-apply + Swish while the input strip is
//   staged (the activated tensor is never written) -> ConvTranspose2d(Cin,1,4,2,1) -> sigmoid -> BCE (+ its gradient)
//   -> input gradient of the layer with the producer's d-Swish and BatchNorm-backward sums -> the layer's weight gradient.
// One launch instead of bn_act + convt_last_fwd + im2col + dense-GEMM data gradient + weight-gradient GEMM.
struct DecLastFusedArgs {
    const bf16* r;           // raw (pre-BN) input, NHWC [>= G*B][IH][IW][Cin]
    BnFinalizeArgs fin;      // BatchNorm of the producer layer over ALL its groups; block (0,0) writes tables + running stats
    int act;
    const float* w;          // fp32 (Cin, 1, 4, 4)
    int G, B, IH, IW, Cin;   // G = groups (passes) whose last layer is computed
    int dbg = 0;             // measurement aid (dec_last_ca_kernel): phases switched off
    int Cout = 1;            // output channels (weights (Cin, Cout, 4, 4); targets / dumps NCHW [..][Cout][2IH][2IW])
    int bwd_groups;          // groups [0, bwd_groups) also produce db / red / weight-gradient partials (0: forward only)
    const float* target;     // NCHW [B][1][2IH][2IW] or null
    float* logits; float* recon; float* dlogit;      // optional dumps, NCHW [G*B][1][2IH][2IW]
    float coef[4];
    float* loss_sum;         // [MMVAE_LOSS_SLOTS][16] column g += BCE sums, or null
    bf16* db;                // out NHWC [G*B][IH][IW][Cin]
    float2* red;             // [fin.G][MMVAE_STAT_SLOTS][Cin] += (sum db, sum db*xhat)
    float* wslab;            // [bwd_groups*B*strips][Cin][16]: per-workgroup weight-gradient partials (plain stores)
};
int dec_last_fused_strips(int IH);
// matrix-core form (dec_last.hip): one workgroup per image, i.e. ONE weight-gradient partial per image (strips = 1)
bool dec_last_mfma_applies(const DecLastFusedArgs& a);
int launch_dec_last_mfma(const DecLastFusedArgs& a, hipStream_t s);
int launch_dec_last_fused(const DecLastFusedArgs& a, hipStream_t s);
// CelebA tail (dec_last.hip): 32x32x32 -> 64x64x3, one workgroup per image in 4 strips; wslab [bwd_groups*B][32][48], column
// j = tap*3 + co (the packed layout of the layer's gradient, plan_base.h build_conv thin_out)
bool dec_last_ca_applies(const DecLastFusedArgs& a);
int launch_dec_last_ca(const DecLastFusedArgs& a, hipStream_t s);

// First encoder layer of MultiMNIST on the matrix cores (conv1.hip): Conv2d(1, 32, 4, 2, 1) on 50x50 images, raw + Swish
// copies [B][25][25][32] bf16; and its weight gradient added (float atomics) into the packed gradient [32][Kpad].
int launch_conv1_fwd_mfma(const float* image, int B, const bf16* Wp, int Kpad, bf16* r1, bf16* a1, hipStream_t s);
int launch_conv1_wgrad_mfma(const float* image, int B, const bf16* d1, float* dWp, int Kpad, hipStream_t s);
