// CelebA MMVAE plan (celeba/model.py:14-57,91-196 ; celeba/train.py:60-81,131-147).
#pragma once
#include "layers.h"

struct PlanBase;

struct CelebaStepIO {
    void* ws = nullptr; size_t ws_bytes = 0;   // caller-owned scratch
    const long long* step_ctr = nullptr;       // device step counter keying the Philox streams (may be null)
    const float* image = nullptr;       // [B][3][64][64] fp32
    const float* attrs = nullptr;       // [B][18] fp32 (0/1)
    const float* eps = nullptr;         // [3][B][D] injected N(0,1) draws, or null -> Philox
    const uint8_t* enc_mask = nullptr;  // [2][B][1024] keep flags of classifier Dropout(0.1), or null -> Philox
    int enc_dropout = 1;                // 0 disables the dropout (fixtures with p=0)
    float kl_lambda = 1e-3f;            // celeba/train.py:61 default, never overridden by train()
    float lambda_x[3] = {1.f, 1.f, 1.f};
    float lambda_y[3] = {1.f, 1.f, 1.f};
    unsigned long long seed = 0x243F6A8885A308D3ull;
    // outputs
    float* sums = nullptr;              // [16]: image bce_sum[0..2], attrs bce_sum[4..6], kl_sum[8..10]
    float* recon_image = nullptr;       // [3][B][3][64][64] or null
    float* recon_attrs = nullptr;       // [3][B][18] or null
    float* mu = nullptr; float* logvar = nullptr;   // [3][B][D] or null
    int pass_skip[3] = {0, 0, 0};       // 1: pass k is absent from this step
    int defer_unpack = 0;               // 1: leave the GEMM-weight gradients packed (the optimizer gathers them: grad_map)
};

struct CelebaPlan;
CelebaPlan* celeba_create(int D, int B);
void celeba_destroy(CelebaPlan*);
PlanBase* celeba_base(CelebaPlan*);
int celeba_step(CelebaPlan*, const CelebaStepIO&, int training, int do_backward, hipStream_t);
// granular module entry points (drop-in nn.Module forwards); B rows, every call brings its workspace
int celeba_image_encoder_fwd(CelebaPlan*, void* ws, size_t wsb, const float* image, const uint8_t* mask, int training, float* out, hipStream_t);
int celeba_image_encoder_bwd(CelebaPlan*, void* ws, size_t wsb, const float* d_out, const uint8_t* mask, hipStream_t);
int celeba_image_decoder_fwd(CelebaPlan*, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t);
int celeba_image_decoder_bwd(CelebaPlan*, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t);
int celeba_attrs_encoder_fwd(CelebaPlan*, void* ws, size_t wsb, const float* attrs, int training, float* out, hipStream_t);
int celeba_attrs_encoder_bwd(CelebaPlan*, void* ws, size_t wsb, const float* d_out, hipStream_t);
int celeba_attrs_decoder_fwd(CelebaPlan*, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t);
int celeba_attrs_decoder_bwd(CelebaPlan*, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t);
