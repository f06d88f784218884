// MultiMNIST MMVAE plan (multimnist/model.py:21-93,150-307 ; multimnist/train.py:69-87,146-173).
#pragma once
#include "layers.h"
#include "text.h"

struct ModelBuffers;
typedef ModelBuffers MMBuffers;

struct MMStepIO {
    void* ws = nullptr; size_t ws_bytes = 0;   // caller-owned scratch (mm_workspace_bytes)
    const long long* step_ctr = nullptr;       // device step counter keying the Philox streams (may be null)
    const float* image = nullptr;       // [B][1][50][50] fp32
    const long long* text = nullptr;    // [B][4] int64
    const float* eps = nullptr;         // [3][B][D] injected N(0,1) draws, or null -> Philox
    const uint8_t* enc_mask1 = nullptr; // [2][B][400] keep flags (null -> Philox, p=0.1)
    const uint8_t* enc_mask2 = nullptr; // [2][B][200]
    const uint8_t* gru_keep = nullptr;  // [4][3B][100]
    int enc_dropout = 1;                // 0 disables the classifier dropout (fixtures with p=0)
    int gru_dropout = 1;
    const long long* force_tokens = nullptr;   // [3B][4] test hook
    float kl_lambda = 1e-3f;
    float lambda_xy[3] = {1.f, 1.f, 0.f};      // multimnist/train.py:158-166
    float lambda_yx[3] = {1.f, 0.5f, 1.f};
    unsigned long long seed = 0x243F6A8885A308D3ull;
    // outputs
    float* sums = nullptr;              // [16]: bce_sum[0..2], nll_sum[4..6], kl_sum[8..10]
    float* recon_image = nullptr;       // [3][B][2500] or null
    float* recon_text = nullptr;        // [3][B][4][12] or null
    float* mu = nullptr; float* logvar = nullptr;   // [3][B][D] or null
    long long* tokens = nullptr;        // [3][B][4] or null
    int pass_skip[3] = {0, 0, 0};       // 1: pass k is absent from this step (paired_weak.py / modal_weak.py)
    int pack_first = 0;                 // 1: the prologue launch also refreshes the packed bf16 weights (after an optimizer step)
    int defer_unpack = 0;               // 1: leave the GEMM-weight gradients in their packed buffers (the optimizer kernel
                                        // gathers them through mm_grad_map and completes the flat gradient itself)
    int dp_split = 0;                   // 1 (data-parallel step): the decoders' gradients (image_decoder.*, text_decoder.*) are
                                        // unpacked into the flat buffer as soon as they are complete -- mm_wait_early_grads --
                                        // so that their all-reduce overlaps the encoders' backward; the rest at the end as usual
    // the optimizer update of the decoders' parameters inside the step (include/mmvae_hip.h: mmvae_early_adam)
    bool early_adam = false;
    float* ea_m = nullptr; float* ea_v = nullptr; long long* ea_state = nullptr;
    float ea_lr = 0.f, ea_b1 = 0.f, ea_b2 = 0.f, ea_eps = 0.f, ea_scale = 1.f;
    const int* ea_gmap = nullptr; int* ea_ran = nullptr;
};

struct MMPlan;
// (offset, length) runs of image_decoder.* / text_decoder.* in the flat buffers; returns their number (<= cap)
int mm_early_ranges(const MMPlan* P, long long* ranges, int cap);
MMPlan* mm_create(int D, int B);
void mm_destroy(MMPlan*);
int mm_D(const MMPlan*);
int mm_B(const MMPlan*);
const std::vector<ParamInfo>& mm_params(const MMPlan*);
long long mm_param_count(const MMPlan*);
long long mm_packed_elems(const MMPlan*);
long long mm_packed_vec_elems(const MMPlan*);
long long mm_gpk_elems(const MMPlan*);
long long mm_gpk_vec_elems(const MMPlan*);
int mm_ndesc(const MMPlan*); const PackDesc* mm_desc_host(const MMPlan*);
int mm_ngdesc(const MMPlan*); const PackDesc* mm_gdesc_host(const MMPlan*);
size_t mm_workspace_bytes(const MMPlan*);
size_t mm_module_workspace_bytes(const MMPlan*);
int mm_bind(MMPlan*, const MMBuffers&);
int mm_pack_weights(MMPlan*, hipStream_t);
int mm_grad_map(MMPlan*, int* map, hipStream_t);      // [param_count] see AdamArgs::gmap
// forward (3 passes) + losses; training!=0 also runs backward into `grads` (which the caller zeroed)
int mm_step_fwd_bwd(MMPlan*, const MMStepIO&, int training, int do_backward, hipStream_t);
// granular module entry points (drop-in modules); every call brings its own workspace
int mm_image_encoder_fwd(MMPlan*, void* ws, size_t wsb, const float* image, const uint8_t* m1, const uint8_t* m2,
                         int training, float* out, hipStream_t);
int mm_image_encoder_bwd(MMPlan*, void* ws, size_t wsb, const float* d_out, const uint8_t* m1, const uint8_t* m2, hipStream_t);
int mm_image_decoder_fwd(MMPlan*, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t);
int mm_image_decoder_bwd(MMPlan*, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t);
int mm_text_encoder_fwd(MMPlan*, void* ws, size_t wsb, const long long* text, float* out, hipStream_t);
int mm_text_encoder_bwd(MMPlan*, void* ws, size_t wsb, const long long* text, const float* d_out, hipStream_t);
int mm_text_decoder_fwd(MMPlan*, void* ws, size_t wsb, const float* z, int training, const uint8_t* keep,
                        const long long* force_tokens, float* words, long long* tokens, hipStream_t);
int mm_text_decoder_bwd(MMPlan*, void* ws, size_t wsb, const float* z, const uint8_t* keep, const long long* force_tokens,
                        const float* words, const long long* tokens, const float* d_words, float* dz, hipStream_t);
int mm_unpack_grads(MMPlan*, hipStream_t);
// makes `s` wait until the early gradient part of the last dp_split step is complete in the flat gradient buffer
int mm_wait_early_grads(MMPlan*, hipStream_t s);
int mm_bench_layer(MMPlan*, void* ws, size_t wsb, const char* layer, int iters, hipStream_t);
double mm_layer_flops(const MMPlan*, const char* layer);
double mm_layer_algo_flops(const MMPlan*, const char* layer);
double mm_layer_algo_bytes(const MMPlan*, const char* layer);
int mm_num_bn(const MMPlan*);
int mm_bn_info(const MMPlan*, int i, std::string& prefix, int& C, long long& offset);
long long mm_bn_floats(const MMPlan*);
long long mm_debug_offset(MMPlan*, const char* name);
