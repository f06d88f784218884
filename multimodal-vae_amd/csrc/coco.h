// COCO MMVAE plan (coco/model.py:22-90,147-312 ; coco/train.py:66-84,146-165).
#pragma once
#include "layers.h"

struct PlanBase;

struct CocoStepIO {
    void* ws = nullptr; size_t ws_bytes = 0;   // caller-owned scratch
    const long long* step_ctr = nullptr;       // device step counter keying the Philox streams (may be null)
    const float* image = nullptr;       // [B][3][32][32] fp32
    const float* text = nullptr;        // [B][T][300] fp32 (GloVe vectors, zero rows after the caption)
    const float* sos = nullptr;         // [300]: GloVe('<s>'), the decoder's first input (coco/model.py:271-272)
    const float* eps = nullptr;         // [3][B][D] injected N(0,1) draws, or null -> Philox
    const uint8_t* enc_mask1 = nullptr; // [2][B][1024] keep flags of classifier Dropout 1, or null -> Philox
    const uint8_t* enc_mask2 = nullptr; // [2][B][256]
    const uint8_t* gru_keep = nullptr;  // [T][3B][200] keep flags of the decoder GRU's inter-layer dropout, or null -> Philox
    int enc_dropout = 1;                // 0 disables the dropouts (fixtures with p = 0)
    int gru_dropout = 1;
    float kl_lambda = 1e-3f;
    float lambda_xy[3] = {1.f, 1.f, 0.f};      // coco/train.py:152-164
    float lambda_yx[3] = {1.f, 1.f, 1.f};
    unsigned long long seed = 0x243F6A8885A308D3ull;
    // outputs
    float* sums = nullptr;              // [16]: image bce_sum[0..2], text squared-error sum[4..6], kl_sum[8..10]
    float* recon_image = nullptr;       // [3][B][3][32][32] or null
    float* recon_text = nullptr;        // [3][B][T][300] or null
    float* mu = nullptr; float* logvar = nullptr;   // [3][B][D] or null
    int pass_skip[3] = {0, 0, 0};       // 1: pass k is absent from this step
    int pack_first = 0;                 // 1: the step refreshes the packed bf16 weights itself (after an optimizer step): the caption
                                        //    GRUs' part on the text stream, the image half's on the main stream, side by side
    int defer_unpack = 0;               // 1: leave the GEMM-weight gradients packed (mmvae_adam_step_packed gathers them)
    long long* optimizer_state = nullptr;   // Adam's 16-byte state block: skip word set by a step that timed out (plan_base.h sum_slots_kernel)
};

struct CocoPlan;
CocoPlan* coco_create(int D, int B, int T);      // T: caption length (coco/utils.py:12-15: 102)
void coco_destroy(CocoPlan*);
PlanBase* coco_base(CocoPlan*);
int coco_steps(const CocoPlan*);
int coco_step(CocoPlan*, const CocoStepIO&, int training, int do_backward, hipStream_t);
// granular module entry points (drop-in nn.Module forwards); B rows, every call brings its workspace
int coco_image_encoder_fwd(CocoPlan*, void* ws, size_t wsb, const float* image, const uint8_t* m1, const uint8_t* m2, int training, float* out, hipStream_t);
int coco_image_encoder_bwd(CocoPlan*, void* ws, size_t wsb, const float* d_out, const uint8_t* m1, const uint8_t* m2, hipStream_t);
int coco_image_decoder_fwd(CocoPlan*, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t);
int coco_image_decoder_bwd(CocoPlan*, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t);
int coco_text_encoder_fwd(CocoPlan*, void* ws, size_t wsb, const float* text, float* out, hipStream_t);
int coco_text_encoder_bwd(CocoPlan*, void* ws, size_t wsb, const float* text, const float* d_out, hipStream_t);
int coco_text_decoder_fwd(CocoPlan*, void* ws, size_t wsb, const float* z, const float* sos, const uint8_t* keep, int training, float* sentence, hipStream_t);
int coco_text_decoder_bwd(CocoPlan*, void* ws, size_t wsb, const float* z, const float* sos, const uint8_t* keep, const float* sentence, const float* d_sentence, float* dz, hipStream_t);
