// MNIST MMVAE plan (mnist/model.py:14-185 ; mnist/train.py:64-81,131-147).
#pragma once
#include "layers.h"

struct PlanBase;

struct MnistStepIO {
    void* ws = nullptr; size_t ws_bytes = 0;   // caller-owned scratch (mnist_workspace_bytes)
    const long long* step_ctr = nullptr;       // device step counter keying the Philox stream (may be null)
    const float* image = nullptr;       // [B][1][28][28] fp32 (viewed as [B][784], mnist/model.py:114)
    const long long* label = nullptr;   // [B] int64 digit labels
    const float* eps = nullptr;         // [3][B][D] injected N(0,1) draws, or null -> Philox
    float lambda_xy[3] = {1.f, 1.f, 1.f};      // mnist/train.py:137-146 (defaults of loss_function)
    float lambda_yx[3] = {1.f, 1.f, 1.f};
    float kl_coef = 0.f;                // d loss / d kl_sum = 1 / (B * 784/3)   (mnist/train.py:78-79)
    unsigned long long seed = 0x243F6A8885A308D3ull;
    // outputs
    float* sums = nullptr;              // [16]: bce_sum[0..2], nll_sum[4..6], kl_sum[8..10]
    float* recon_image = nullptr;       // [3][B][784] or null
    float* recon_text = nullptr;        // [3][B][10] log-probs or null
    float* mu = nullptr; float* logvar = nullptr;   // [3][B][D] or null
    int pass_skip[3] = {0, 0, 0};       // 1: pass k is absent from this step (mnist/paired_weak.py, mnist/modal_weak.py)
};

struct MnistPlan;
MnistPlan* mnist_create(int D, int B, int precision = -1);   // 0 fp32 (default), 1 bf16 operands, -1: env MMVAE_MNIST_PRECISION
int mnist_is_f32(const MnistPlan*);
void mnist_destroy(MnistPlan*);
PlanBase* mnist_base(MnistPlan*);
int mnist_step(MnistPlan*, const MnistStepIO&, int training, int do_backward, hipStream_t);
// granular module entry points (drop-in nn.Module forwards); B rows, every call brings its workspace
int mnist_image_encoder_fwd(MnistPlan*, void* ws, size_t wsb, const float* image, int training, float* out, hipStream_t);
int mnist_image_encoder_bwd(MnistPlan*, void* ws, size_t wsb, const float* d_out, hipStream_t);
int mnist_image_decoder_fwd(MnistPlan*, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t);
int mnist_image_decoder_bwd(MnistPlan*, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t);
int mnist_text_encoder_fwd(MnistPlan*, void* ws, size_t wsb, const long long* label, int training, float* out, hipStream_t);
int mnist_text_encoder_bwd(MnistPlan*, void* ws, size_t wsb, const long long* label, const float* d_out, hipStream_t);
int mnist_text_decoder_fwd(MnistPlan*, void* ws, size_t wsb, const float* z, int training, float* logp, hipStream_t);
int mnist_text_decoder_bwd(MnistPlan*, void* ws, size_t wsb, const float* d_logp, const float* logp, float* dz, hipStream_t);
