// Private: the MNIST plan object shared by mnist.hip (bf16-MFMA path) and mnist_f32.hip (fp32 path).
#pragma once
#include "mnist.h"
#include "plan_base.h"

struct MnistPlan : PlanBase {
    int ldz;
    MlpLin ie[3], id[3], te_lin, td[2];
    BnL bn[6];
    long long emb_off;
    struct W {
        char* zero_begin; size_t zero_bytes;
        float2* st[6]; float2* red[6];
        float* sums; float* dz_img; float* dz_txt;
        float2* aff[6]; float2* mr[6];
        bf16 *x_bf, *r_ie[2], *a_ie[2]; float* encout;
        bf16 *r_te, *a_te; float* txtout;
        float *eps, *mu, *logvar, *z_f32; bf16* z_bf;
        bf16 *r_id[2], *a_id[2]; float *logits, *dlogit; bf16* dlogit_bf;
        bf16 *r_td, *a_td; float *tlogits, *words; bf16* dtl;
        bf16 *d_id[2], *d_td;
        bf16 *d_encout, *d_txtout_bf, *d_ie[2], *d_te;
    } w;
    // ---- fp32 path (mnist_f32.hip): precision of the reference itself; the default for this model family
    bool f32 = true;
    struct W32 {
        float* sums; float* dz_img; float* dz_txt; char* zero_begin; size_t zero_bytes;
        float2* mr[6];                                   // [G][C] (mean, rstd) of every BatchNorm
        float *r_ie[2], *a_ie[2], *encout;
        float *r_te, *a_te, *txtout;
        float *eps, *mu, *logvar, *z; bf16* z_bf;
        float *r_id[2], *a_id[2], *logits, *dlogit;
        float *r_td, *a_td, *tlogits, *words, *dtl;
        float *d_id[2], *d_td, *d_encout, *d_txtout, *d_ie[2], *d_te;
    } w32;
};


// fp32 path entry points (mnist_f32.hip); same contracts as the public functions of mnist.h
size_t mnist_f32_workspace_bytes(MnistPlan& P);
int mnist_f32_step(MnistPlan& P, const MnistStepIO& io, int training, int do_backward, hipStream_t s);
int mnist_f32_image_encoder_fwd(MnistPlan& P, void* ws, size_t wsb, const float* image, int training, float* out, hipStream_t s);
int mnist_f32_image_encoder_bwd(MnistPlan& P, void* ws, size_t wsb, const float* d_out, hipStream_t s);
int mnist_f32_image_decoder_fwd(MnistPlan& P, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s);
int mnist_f32_image_decoder_bwd(MnistPlan& P, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s);
int mnist_f32_text_encoder_fwd(MnistPlan& P, void* ws, size_t wsb, const long long* label, int training, float* out, hipStream_t s);
int mnist_f32_text_encoder_bwd(MnistPlan& P, void* ws, size_t wsb, const long long* label, const float* d_out, hipStream_t s);
int mnist_f32_text_decoder_fwd(MnistPlan& P, void* ws, size_t wsb, const float* z, int training, float* logp, hipStream_t s);
int mnist_f32_text_decoder_bwd(MnistPlan& P, void* ws, size_t wsb, const float* d_logp, const float* logp, float* dz, hipStream_t s);
