/* libmmvae_hip.so -- C-ABI of the MI355X (gfx950) MMVAE ELBO engine.
 *
 * The reference (wenxuanliu/multimodal-vae) has no FFI: its hot path is reached through the Python module surface
 * of <ds>/model.py and loss_function in <ds>/train.py.  Each entry point below names the reference code it
 * replaces (paths under the reference repo).  Conventions:
 *   - plain C types only; every device buffer is owned by the caller (e.g. tensor.data_ptr()); the library never
 *     allocates or frees device memory; scratch comes from a caller-provided workspace whose size is queried;
 *   - all work is enqueued on the caller's hipStream_t (passed as void*), no hidden synchronisation;
 *   - returns 0 on success or a negative MMVAE_E* code; mmvae_last_error() gives a thread-local message;
 *   - plans (mmvae_mm_t) are host-side descriptors: one host thread per plan, one process per GPU.
 */
#ifndef MMVAE_HIP_H
#define MMVAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMVAE_OK 0
#define MMVAE_EINVAL (-1)
#define MMVAE_EHIP (-2)
#define MMVAE_ENOSPC (-3)
#define MMVAE_ESTATE (-4)
#define MMVAE_ETIMEOUT (-5)               /* a device-side exchange of the step gave up: the step is void (mmvae_step_status) */

int mmvae_init(int device);                 /* checks the device is gfx950, makes it current */
const char* mmvae_last_error(void);
const char* mmvae_version(void);
/* Priority policy of the engine's three side streams (second-modality path, weight gradients); call before the first
 * step of the process.  1 (the default): default priority.  0: lowest priority, the side streams only fill CUs the main
 * chain leaves idle -- up to 0.8 % faster (CelebA), but ONLY while no other stream of the process carries GPU work: with
 * an H2D copy stream or a collective library's stream (RCCL under torch.distributed) active, mixed priorities slow every
 * kernel down 2-3x.  MMVAE_ESTATE if the side streams already exist with the other policy. */
int mmvae_set_stream_policy(int flat);

/* ---- collective face: the path's ONE exchange step, for hosts without a communicator of their own ----------
 * In-place SUM all-reduce of the flat fp32 gradient buffer over RCCL / xGMI, enqueued on the caller's stream; fold the
 * 1/world averaging into mmvae_adam_step's grad_scale.  librccl is bound with dlopen on first use (a copy already loaded
 * in the process -- PyTorch's -- is reused); nothing else in the library depends on it.  Rank 0 creates the 128-byte id
 * (ncclUniqueId) and hands it to the other ranks out of band (file, socket, MPI); every rank then calls mmvae_comm_init
 * with its HIP device already selected.  A PyTorch host keeps torch.distributed. */
typedef struct mmvae_comm mmvae_comm_t;
int mmvae_comm_unique_id(void* out128);
int mmvae_comm_init(mmvae_comm_t** out, int rank, int world, const void* unique_id128);
int mmvae_allreduce_grads(mmvae_comm_t*, float* flat, size_t n, void* stream);
int mmvae_comm_world(const mmvae_comm_t*);
int mmvae_comm_destroy(mmvae_comm_t*);

/* ---------------------------------------------------------------- MultiMNIST plan (multimnist/model.py:21-93) */
typedef struct MMPlan mmvae_mm_t;
mmvae_mm_t* mmvae_mm_create(int n_latents, int batch);      /* MultimodalVAE(n_latents) at a fixed batch size */
void mmvae_mm_destroy(mmvae_mm_t*);
long long mmvae_mm_param_count(const mmvae_mm_t*);          /* scalars in the flat fp32 parameter buffer */
int mmvae_mm_num_params(const mmvae_mm_t*);                 /* tensors, state_dict order (52) */
/* name (<=127 chars), ndim, shape[4], element offset of parameter i */
int mmvae_mm_param_info(const mmvae_mm_t*, int i, char* name, int* ndim, int* shape, long long* offset);
long long mmvae_mm_bn_floats(const mmvae_mm_t*);            /* running_mean|running_var of every BatchNorm */
int mmvae_mm_num_bn(const mmvae_mm_t*);
int mmvae_mm_bn_info(const mmvae_mm_t*, int i, char* prefix, int* channels, long long* offset);
long long mmvae_mm_packed_elems(const mmvae_mm_t*);         /* bf16 elements of the packed weights */
long long mmvae_mm_packed_vec_elems(const mmvae_mm_t*);     /* fp32 */
long long mmvae_mm_gpk_elems(const mmvae_mm_t*);            /* fp32 elements of the packed weight gradients */
long long mmvae_mm_gpk_vec_elems(const mmvae_mm_t*);
size_t mmvae_mm_desc_bytes(const mmvae_mm_t*, int which);   /* which: 0 = weight table, 1 = gradient table */
int mmvae_mm_desc_copy(const mmvae_mm_t*, int which, void* host_out);   /* caller uploads it to the device */
size_t mmvae_mm_workspace_bytes(const mmvae_mm_t*);
/* workspace of ONE granular module call (the *_encoder_* / *_decoder_* entry points below run a single pass over B rows;
   the fused step batches 3 passes and needs mmvae_mm_workspace_bytes) */
size_t mmvae_mm_module_workspace_bytes(const mmvae_mm_t*);
int mmvae_mm_bind(mmvae_mm_t*, float* params, float* grads, float* bn_stats, long long* bn_num_batches_tracked,
                  void* packed_bf16, float* packed_vec, float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev);
int mmvae_mm_pack_weights(mmvae_mm_t*, void* stream);    /* refresh bf16 GEMM-layout copies after params change */
/* map[i] (int32, param_count entries) = where flat parameter i's packed gradient lives: >= 0 index into gpk, <= -2:
 * -(index + 2) into gpk_vec, -1 none.  Built once per parameter layout; input of mmvae_adam_step_packed. */
int mmvae_mm_grad_map(mmvae_mm_t*, int* map, void* stream);

/* Optimizer arguments of mmvae_mm_step_io.early_adam (torch.optim.Adam semantics, as mmvae_adam_step_packed). */
struct mmvae_early_adam {
    float* m; float* v;                     /* Adam moments, flat like the parameters */
    long long* state;                       /* the 16-byte optimizer state block (mmvae_adam_step) */
    float lr, beta1, beta2, eps, grad_scale;
    const int* gmap;                        /* mmvae_mm_grad_map */
    int* ran;                               /* host int: set to 1 when the step issued the early part, 0 when it did not (serial mode, a step
                                             * without an image-decoder backward, ...): the caller then updates every range itself */
};
/* One 3-pass ELBO step (multimnist/train.py:150-168): forward of (image,text), (image), (text), the three
 * loss_function sums and -- if do_backward -- the gradient of loss_1+loss_2+loss_3 written to `grads`
 * (the step zeroes `grads` first, i.e. it includes optimizer.zero_grad()). */
typedef struct {
    void* ws; size_t ws_bytes;
    const long long* step_counter;          /* device int64 keying the RNG streams, may be NULL */
    const float* image;                     /* [B][1][50][50] */
    const long long* text;                  /* [B][4] */
    const float* eps;                       /* [3][B][D] or NULL (Philox) */
    const uint8_t* enc_mask1;               /* [2][B][400] keep flags or NULL */
    const uint8_t* enc_mask2;               /* [2][B][200] keep flags or NULL */
    const uint8_t* gru_keep;                /* [4][3B][100] keep flags or NULL */
    int enc_dropout, gru_dropout;           /* 0 turns the respective dropout off (p = 0 fixtures) */
    const long long* force_tokens;          /* [3B][4] or NULL: overrides the greedy feedback (test hook) */
    float kl_lambda;
    float lambda_xy[3], lambda_yx[3];
    unsigned long long seed;
    float* sums;                            /* out [16]: [0..2] BCE sums, [4..6] NLL sums, [8..10] KL sums */
    float* recon_image;                     /* out [3][B][2500] or NULL */
    float* recon_text;                      /* out [3][B][4][12] or NULL */
    float* mu; float* logvar;               /* out [3][B][D] or NULL */
    long long* tokens;                      /* out [3][B][4] or NULL */
    int pass_skip[3];                       /* 1: pass k absent from this step (multimnist/paired_weak.py:84-117,
                                             * modal_weak.py:87-117): no loss, no gradient, no BatchNorm running update */
    int defer_unpack;                       /* 1: GEMM-weight gradients stay in the packed buffers for mmvae_adam_step_packed
                                             * (loss.backward() + optimizer.step() of multimnist/train.py:168,173 in one pass) */
    int pack_first;                         /* 1: the step's prologue launch also refreshes the packed bf16 weights from the
                                             * parameters (what mmvae_mm_pack_weights does): set it after an optimizer step
                                             * instead of calling mmvae_mm_pack_weights -- one launch less on the step's chain */
    int dp_split;                           /* 1 (data-parallel replica): the gradients of image_decoder.* and text_decoder.* are
                                             * complete in `grads` before the encoders' backward has run -- order a communication
                                             * stream behind them with mmvae_mm_wait_early_grads and all-reduce those two parameter
                                             * ranges while the rest of the step runs; the other ranges are complete when the step is */
    const struct mmvae_early_adam* early_adam;  /* NULL, or (with defer_unpack = 1, no data parallelism): the optimizer update of image_decoder.* and
                                             * text_decoder.* is issued INSIDE the step, on the weight-gradient stream, as soon as their gradients are
                                             * complete -- it runs beside the encoders' backward instead of behind the whole step.  The caller finishes
                                             * the optimizer step with mmvae_adam_step_packed_ranges over the remaining ranges when *ran was set. */
} mmvae_mm_step_io;
int mmvae_mm_step(mmvae_mm_t*, const mmvae_mm_step_io*, int training, int do_backward, void* stream);
/* `stream` waits (hipStreamWaitEvent) for the early gradient part of the most recent dp_split step of this plan.
 * The collective itself stays with the host framework (torch.distributed over RCCL): INTEGRATION.md, Data parallelism. */
int mmvae_mm_wait_early_grads(mmvae_mm_t*, void* stream);

/* Granular modules (drop-in for ImageEncoder/ImageDecoder/TextEncoder/TextDecoder.forward + autograd backward).
 * Every forward keeps its saved activations in the workspace passed to it; pass the same one to the backward. */
int mmvae_mm_image_encoder_fwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const float* image, const uint8_t* mask1,
                               const uint8_t* mask2, int training, float* out_mu_logvar, void* stream);   /* model.py:183-188 */
int mmvae_mm_image_encoder_bwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const float* d_out, const uint8_t* mask1,
                               const uint8_t* mask2, void* stream);
int mmvae_mm_image_decoder_fwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const float* z, int training, float* recon,
                               void* stream);                                                           /* model.py:211-216 */
int mmvae_mm_image_decoder_bwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const float* d_recon, const float* recon,
                               float* dz, void* stream);
int mmvae_mm_text_encoder_fwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const long long* text, float* out_mu_logvar,
                              void* stream);                                                            /* model.py:237-247 */
int mmvae_mm_text_encoder_bwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const long long* text, const float* d_out,
                              void* stream);
int mmvae_mm_text_decoder_fwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const float* z, int training, const uint8_t* keep,
                              const long long* force_tokens, float* words, long long* tokens, void* stream); /* model.py:268-288 */
int mmvae_mm_text_decoder_bwd(mmvae_mm_t*, void* ws, size_t ws_bytes, const float* z, const uint8_t* keep,
                              const long long* force_tokens, const float* words, const long long* tokens,
                              const float* d_words, float* dz, void* stream);
/* Runs one named GEMM of the step `iters` times on the workspace contents of the last step (profiling aid). */
int mmvae_mm_bench_layer(mmvae_mm_t*, void* ws, size_t ws_bytes, const char* layer, int iters, void* stream);
double mmvae_mm_layer_flops(const mmvae_mm_t*, const char* layer);        /* executed: 2*rows*N*K, zero-padded taps included */
double mmvae_mm_layer_algo_flops(const mmvae_mm_t*, const char* layer);   /* algorithmic: the FlopCounterMode count of the reference layer */
double mmvae_mm_layer_algo_bytes(const mmvae_mm_t*, const char* layer);   /* algorithmic: each operand read once, the result written once (0: not a conv layer) */
/* measurement aid: GEMM FLOPs enqueued by this process since the last reset (counted on the host by the launchers) */
double mmvae_debug_flops(int reset);
/* test / A-B aid: named integer switches read by the launchers ("convres" = 0 turns the image-resident conv kernels off) */
int mmvae_debug_set(const char* key, int value);
/* test aid: byte offset of a named intermediate inside the workspace (-1 if unknown) */
long long mmvae_mm_debug_offset(mmvae_mm_t*, const char* name);

/* ---------------------------------------------------------------- MNIST (mnist/model.py, mnist/train.py)
 * MultimodalVAE of mnist/model.py:14-50: Linear -> BatchNorm1d -> ReLU stacks (ImageEncoder :99-118, ImageDecoder
 * :121-133, TextEncoder :136-153, TextDecoder :156-170), ProductOfExperts :173-185, loss_function mnist/train.py:64-81.
 * The plan/query/bind/pack functions have the same meaning as their mmvae_mm_* counterparts above. */
typedef struct MnistPlan mmvae_mnist_t;
mmvae_mnist_t* mmvae_mnist_create(int n_latents, int batch);      /* MultimodalVAE(n_latents) mnist/model.py:14-20 */
/* precision 0 (default of mmvae_mnist_create): fp32 operands on fp32 MFMA, fp32 activations -- the reference's own
 * arithmetic (this model is Linear->BatchNorm1d->ReLU: bf16 operand rounding flips ReLU decisions and moves gradients by
 * 10-30 %); precision 1: bf16 MFMA operands like the conv models.  -1: environment MMVAE_MNIST_PRECISION=fp32|bf16 */
mmvae_mnist_t* mmvae_mnist_create_p(int n_latents, int batch, int precision);
int mmvae_mnist_precision(const mmvae_mnist_t*);
void mmvae_mnist_destroy(mmvae_mnist_t*);
long long mmvae_mnist_param_count(const mmvae_mnist_t*);
int mmvae_mnist_num_params(const mmvae_mnist_t*);
int mmvae_mnist_param_info(const mmvae_mnist_t*, int i, char* name128, int* ndim, int* shape4, long long* offset);
long long mmvae_mnist_bn_floats(const mmvae_mnist_t*);
int mmvae_mnist_num_bn(const mmvae_mnist_t*);
int mmvae_mnist_bn_info(const mmvae_mnist_t*, int i, char* prefix128, int* channels, long long* offset);
long long mmvae_mnist_packed_elems(const mmvae_mnist_t*);
long long mmvae_mnist_packed_vec_elems(const mmvae_mnist_t*);
long long mmvae_mnist_gpk_elems(const mmvae_mnist_t*);
long long mmvae_mnist_gpk_vec_elems(const mmvae_mnist_t*);
size_t mmvae_mnist_desc_bytes(const mmvae_mnist_t*, int which);
int mmvae_mnist_desc_copy(const mmvae_mnist_t*, int which, void* host_out);
size_t mmvae_mnist_workspace_bytes(const mmvae_mnist_t*);
/* workspace of ONE granular module call (the *_encoder_* / *_decoder_* entry points below run a single pass over B rows;
   the fused step batches 3 passes and needs mmvae_mnist_workspace_bytes) */
size_t mmvae_mnist_module_workspace_bytes(const mmvae_mnist_t*);
int mmvae_mnist_bind(mmvae_mnist_t*, float* params, float* grads, float* bn_stats, long long* num_batches_tracked,
                     void* packed_bf16, float* packed_vec, float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev);
int mmvae_mnist_pack_weights(mmvae_mnist_t*, void* stream);
int mmvae_mnist_grad_map(mmvae_mnist_t*, int* map, void* stream);                /* as mmvae_mm_grad_map */
/* The train() closure body of mnist/train.py:131-147 (3 passes, 3 losses, backward): same contract as mmvae_mm_step */
typedef struct {
    void* ws; size_t ws_bytes;
    const long long* step_counter;          /* device int64 keying the Philox eps stream, or NULL */
    const float* image;                     /* [B][1][28][28] fp32 */
    const long long* label;                 /* [B] int64 */
    const float* eps;                       /* [3][B][D] or NULL (drawn on device) */
    float lambda_xy[3]; float lambda_yx[3]; /* mnist/train.py:137-146: all 1 */
    float kl_coef;                          /* 1 / (B * 784/3)  (mnist/train.py:78-79) */
    unsigned long long seed;
    float* sums;                            /* out [16]: bce_sum[0..2], nll_sum[4..6], kl_sum[8..10] */
    float* recon_image;                     /* out [3][B][784] or NULL */
    float* recon_text;                      /* out [3][B][10] log-probs or NULL */
    float* mu; float* logvar;               /* out [3][B][D] or NULL */
    int pass_skip[3];                       /* 1: pass k absent from this step (mnist/paired_weak.py, mnist/modal_weak.py) */
} mmvae_mnist_step_io;
int mmvae_mnist_step(mmvae_mnist_t*, const mmvae_mnist_step_io*, int training, int do_backward, void* stream);
/* Granular modules of mnist/model.py (forward + autograd backward); workspace rules as for mmvae_mm_*_fwd/bwd.
 * Parameter gradients accumulate into the bound `grads`. */
int mmvae_mnist_image_encoder_fwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const float* image, int training,
                                  float* out_mu_logvar, void* stream);                          /* mnist/model.py:114-118 */
int mmvae_mnist_image_encoder_bwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const float* d_out, void* stream);
int mmvae_mnist_image_decoder_fwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const float* z, int training, float* recon,
                                  void* stream);                                                /* mnist/model.py:131-133 */
int mmvae_mnist_image_decoder_bwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const float* d_recon, const float* recon,
                                  float* dz, void* stream);
int mmvae_mnist_text_encoder_fwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const long long* label, int training,
                                 float* out_mu_logvar, void* stream);                           /* mnist/model.py:149-153 */
int mmvae_mnist_text_encoder_bwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const long long* label, const float* d_out,
                                 void* stream);
int mmvae_mnist_text_decoder_fwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const float* z, int training, float* log_probs,
                                 void* stream);                                                 /* mnist/model.py:168-170 */
int mmvae_mnist_text_decoder_bwd(mmvae_mnist_t*, void* ws, size_t ws_bytes, const float* d_log_probs, const float* log_probs,
                                 float* dz, void* stream);

/* ---------------------------------------------------------------- CelebA (celeba/model.py, celeba/train.py)
 * MultimodalVAE of celeba/model.py:14-57: conv ImageEncoder :91-128 / ImageDecoder :131-161 on 3x64x64 images,
 * AttributeEncoder :164-178 / AttributeDecoder :181-196 on 18 binary attributes (celeba/datasets.py:26-28),
 * loss_function celeba/train.py:60-81.  Plan/query/bind/pack functions as for mmvae_mm_*. */
typedef struct CelebaPlan mmvae_celeba_t;
mmvae_celeba_t* mmvae_celeba_create(int n_latents, int batch);    /* MultimodalVAE(n_latents) celeba/model.py:15-22 */
void mmvae_celeba_destroy(mmvae_celeba_t*);
long long mmvae_celeba_param_count(const mmvae_celeba_t*);
int mmvae_celeba_num_params(const mmvae_celeba_t*);
int mmvae_celeba_param_info(const mmvae_celeba_t*, int i, char* name128, int* ndim, int* shape4, long long* offset);
long long mmvae_celeba_bn_floats(const mmvae_celeba_t*);
int mmvae_celeba_num_bn(const mmvae_celeba_t*);
int mmvae_celeba_bn_info(const mmvae_celeba_t*, int i, char* prefix128, int* channels, long long* offset);
long long mmvae_celeba_packed_elems(const mmvae_celeba_t*);
long long mmvae_celeba_packed_vec_elems(const mmvae_celeba_t*);
long long mmvae_celeba_gpk_elems(const mmvae_celeba_t*);
long long mmvae_celeba_gpk_vec_elems(const mmvae_celeba_t*);
size_t mmvae_celeba_desc_bytes(const mmvae_celeba_t*, int which);
int mmvae_celeba_desc_copy(const mmvae_celeba_t*, int which, void* host_out);
size_t mmvae_celeba_workspace_bytes(const mmvae_celeba_t*);
/* workspace of ONE granular module call (the *_encoder_* / *_decoder_* entry points below run a single pass over B rows;
   the fused step batches 3 passes and needs mmvae_celeba_workspace_bytes) */
size_t mmvae_celeba_module_workspace_bytes(const mmvae_celeba_t*);
int mmvae_celeba_bind(mmvae_celeba_t*, float* params, float* grads, float* bn_stats, long long* num_batches_tracked,
                      void* packed_bf16, float* packed_vec, float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev);
int mmvae_celeba_pack_weights(mmvae_celeba_t*, void* stream);
int mmvae_celeba_grad_map(mmvae_celeba_t*, int* map, void* stream);                /* as mmvae_mm_grad_map */
/* The train() closure body of celeba/train.py:131-147 (3 passes, 3 losses, backward) */
typedef struct {
    void* ws; size_t ws_bytes;
    const long long* step_counter;          /* device int64 keying the Philox streams, or NULL */
    const float* image;                     /* [B][3][64][64] fp32 */
    const float* attrs;                     /* [B][18] fp32 */
    const float* eps;                       /* [3][B][D] or NULL (drawn on device) */
    const uint8_t* enc_mask;                /* [2][B][1024] keep flags of classifier Dropout(0.1) or NULL (drawn) */
    int enc_dropout;                        /* 0: no dropout */
    float kl_lambda;                        /* celeba/train.py:61 (1e-3) */
    float lambda_x[3]; float lambda_y[3];   /* celeba/train.py:138-147: all 1 */
    unsigned long long seed;
    float* sums;                            /* out [16]: image bce_sum[0..2], attrs bce_sum[4..6], kl_sum[8..10] */
    float* recon_image;                     /* out [3][B][3][64][64] or NULL */
    float* recon_attrs;                     /* out [3][B][18] or NULL */
    float* mu; float* logvar;               /* out [3][B][D] or NULL */
    int pass_skip[3];                       /* 1: pass k absent from this step */
    int defer_unpack;                       /* 1: the optimizer consumes the packed gradients itself (see the MultiMNIST step) */
} mmvae_celeba_step_io;
int mmvae_celeba_step(mmvae_celeba_t*, const mmvae_celeba_step_io*, int training, int do_backward, void* stream);
/* Granular modules (forward + autograd backward), workspace rules as for mmvae_mm_*_fwd/bwd */
int mmvae_celeba_image_encoder_fwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* image, const uint8_t* mask,
                                   int training, float* out_mu_logvar, void* stream);            /* celeba/model.py:124-128 */
int mmvae_celeba_image_encoder_bwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* d_out, const uint8_t* mask,
                                   void* stream);
int mmvae_celeba_image_decoder_fwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* z, int training, float* recon,
                                   void* stream);                                                /* celeba/model.py:157-161 */
int mmvae_celeba_image_decoder_bwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* d_recon, const float* recon,
                                   float* dz, void* stream);
int mmvae_celeba_attrs_encoder_fwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* attrs, int training,
                                   float* out_mu_logvar, void* stream);                          /* celeba/model.py:175-178 */
int mmvae_celeba_attrs_encoder_bwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* d_out, void* stream);
int mmvae_celeba_attrs_decoder_fwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* z, int training, float* recon,
                                   void* stream);                                                /* celeba/model.py:191-196 */
int mmvae_celeba_attrs_decoder_bwd(mmvae_celeba_t*, void* ws, size_t ws_bytes, const float* d_recon, const float* recon,
                                   float* dz, void* stream);

/* ---------------------------------------------------------------- COCO (coco/model.py, coco/train.py)
 * MultimodalVAE of coco/model.py:22-90: conv ImageEncoder :147-187 / ImageDecoder :190-216 on 3x32x32 images (the code's
 * size: Scale(32), coco/train.py:107-112), TextEncoder :219-245 (biGRU over `steps` 300-d GloVe vectors) / TextDecoder
 * :248-312 (2-layer GRU regressing one 300-d vector per step, fed back as the next input), loss_function
 * coco/train.py:66-84 (BCE + MSE + KL).  Plan/query/bind/pack functions as for mmvae_mm_*. */
typedef struct CocoPlan mmvae_coco_t;
mmvae_coco_t* mmvae_coco_create(int n_latents, int batch);        /* MultimodalVAE(n_latents), steps = 102 (coco/utils.py:12-15) */
mmvae_coco_t* mmvae_coco_create_t(int n_latents, int batch, int steps);   /* other caption lengths (tests) */
void mmvae_coco_destroy(mmvae_coco_t*);
int mmvae_coco_steps(const mmvae_coco_t*);
long long mmvae_coco_param_count(const mmvae_coco_t*);
int mmvae_coco_num_params(const mmvae_coco_t*);
int mmvae_coco_param_info(const mmvae_coco_t*, int i, char* name128, int* ndim, int* shape4, long long* offset);
long long mmvae_coco_bn_floats(const mmvae_coco_t*);
int mmvae_coco_num_bn(const mmvae_coco_t*);
int mmvae_coco_bn_info(const mmvae_coco_t*, int i, char* prefix128, int* channels, long long* offset);
long long mmvae_coco_packed_elems(const mmvae_coco_t*);
long long mmvae_coco_packed_vec_elems(const mmvae_coco_t*);
long long mmvae_coco_gpk_elems(const mmvae_coco_t*);
long long mmvae_coco_gpk_vec_elems(const mmvae_coco_t*);
size_t mmvae_coco_desc_bytes(const mmvae_coco_t*, int which);
int mmvae_coco_desc_copy(const mmvae_coco_t*, int which, void* host_out);
size_t mmvae_coco_workspace_bytes(const mmvae_coco_t*);
/* workspace of ONE granular module call (the *_encoder_* / *_decoder_* entry points below run a single pass over B rows;
   the fused step batches 3 passes and needs mmvae_coco_workspace_bytes) */
size_t mmvae_coco_module_workspace_bytes(const mmvae_coco_t*);
int mmvae_coco_bind(mmvae_coco_t*, float* params, float* grads, float* bn_stats, long long* num_batches_tracked,
                    void* packed, float* packed_vec, float* gpk, float* gpk_vec, void* desc_dev, void* gdesc_dev);
int mmvae_coco_pack_weights(mmvae_coco_t*, void* stream);
int mmvae_coco_grad_map(mmvae_coco_t*, int* map, void* stream);                /* as mmvae_mm_grad_map */
/* The train() closure body of coco/train.py:138-173 (3 passes, 3 losses, backward) */
typedef struct {
    void* ws; size_t ws_bytes;
    const long long* step_counter;          /* device int64 keying the Philox streams, or NULL */
    const float* image;                     /* [B][3][32][32] fp32 */
    const float* text;                      /* [B][steps][300] fp32 GloVe vectors (coco/utils.py:36-47) */
    const float* sos;                       /* [300] GloVe('<s>') (coco/model.py:271-272) */
    const float* eps;                       /* [3][B][D] or NULL (drawn on device) */
    const uint8_t* enc_mask1;               /* [2][B][1024] keep flags of classifier Dropout 1 or NULL (drawn) */
    const uint8_t* enc_mask2;               /* [2][B][256] */
    const uint8_t* gru_keep;                /* [steps][3B][200] keep flags of the decoder GRU's inter-layer dropout or NULL */
    int enc_dropout; int gru_dropout;       /* 0: no dropout */
    float kl_lambda;                        /* coco/train.py:67 (1e-3) */
    float lambda_xy[3]; float lambda_yx[3]; /* coco/train.py:152-164: (1,1,0) and (1,1,1) */
    unsigned long long seed;
    float* sums;                            /* out [16]: image bce_sum[0..2], text squared-error sum[4..6], kl_sum[8..10] */
    float* recon_image;                     /* out [3][B][3][32][32] or NULL */
    float* recon_text;                      /* out [3][B][steps][300] or NULL */
    float* mu; float* logvar;               /* out [3][B][D] or NULL */
    int pass_skip[3];                       /* 1: pass k absent from this step */
    int defer_unpack;                       /* 1: the optimizer consumes the packed gradients itself (see the MultiMNIST step) */
    int pack_first;                         /* 1: the step refreshes the packed bf16 weights itself (the caller skipped mmvae_coco_pack_weights
                                               after the optimizer step): caption half on the text stream, image half on the main one */
    long long* optimizer_state;             /* the 16-byte state block this step's mmvae_adam_step* call will get, or NULL.  The caption
                                               decoder's workgroups exchange hidden-state slices through memory and give up after a
                                               bounded spin (a device that cannot keep them co-resident); such a step is void: its `sums`
                                               are NaN, its gradient is marked (element 0 = the NaN 0x7fc0dead, whose payload survives a SUM all-reduce) and
                                               the skip word of this block is set -- mmvae_adam_step* then leaves parameters, moments
                                               and step count untouched.  mmvae_step_status reports it to the host */
} mmvae_coco_step_io;
int mmvae_coco_step(mmvae_coco_t*, const mmvae_coco_step_io*, int training, int do_backward, void* stream);
/* Granular modules (forward + autograd backward), workspace rules as for mmvae_mm_*_fwd/bwd */
int mmvae_coco_image_encoder_fwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* image, const uint8_t* mask1,
                                 const uint8_t* mask2, int training, float* out_mu_logvar, void* stream);  /* coco/model.py:182-187 */
int mmvae_coco_image_encoder_bwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* d_out, const uint8_t* mask1,
                                 const uint8_t* mask2, void* stream);
int mmvae_coco_image_decoder_fwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* z, int training, float* recon,
                                 void* stream);                                                  /* coco/model.py:211-216 */
int mmvae_coco_image_decoder_bwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* d_recon, const float* recon,
                                 float* dz, void* stream);
int mmvae_coco_text_encoder_fwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* text, float* out_mu_logvar,
                                void* stream);                                                   /* coco/model.py:236-245 */
int mmvae_coco_text_encoder_bwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* text, const float* d_out, void* stream);
int mmvae_coco_text_decoder_fwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* z, const float* sos,
                                const uint8_t* keep, int training, float* sentence, void* stream);  /* coco/model.py:266-288 */
int mmvae_coco_text_decoder_bwd(mmvae_coco_t*, void* ws, size_t ws_bytes, const float* z, const float* sos,
                                const uint8_t* keep, const float* sentence, const float* d_sentence, float* dz, void* stream);

/* ---------------------------------------------------------------- dataset-independent ops */
/* ProductOfExperts.forward (multimnist/model.py:355-360) over M stacked experts of n scalars each */
int mmvae_poe_fwd(const float* mu, const float* logvar, int M, int n, float* out_mu, float* out_logvar, void* stream);
int mmvae_poe_bwd(const float* mu, const float* logvar, int M, int n, const float* g_mu, const float* g_logvar,
                  float* d_mu, float* d_logvar, void* stream);
/* MultimodalVAE.reparametrize (multimnist/model.py:33-39), train mode */
int mmvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, int n, float* z, void* stream);
int mmvae_reparam_bwd(const float* logvar, const float* eps, const float* dz, int n, float* d_mu, float* d_logvar, void* stream);
/* KL term of loss_function (multimnist/train.py:85): out[0] += -0.5*sum(1+lv-mu^2-exp(lv)) */
int mmvae_kl_fwd(const float* mu, const float* logvar, int n, float* out_sum, void* stream);
/* every *_bwd below: the gradient is coef * (*gscale) * d(sum)/d(input); gscale (nullable) is a DEVICE scalar -- the upstream
   gradient of the 0-d loss tensor, read by the kernel so that the host never synchronises on it */
int mmvae_kl_bwd(const float* mu, const float* logvar, int n, float coef, const float* gscale, float* d_mu, float* d_logvar, void* stream);
/* F.binary_cross_entropy on probabilities (multimnist/train.py:75): out[0] += sum of terms; bwd = coef * d/dp */
int mmvae_bce_fwd(const float* p, const float* target, long long n, float* out_sum, void* stream);
int mmvae_bce_bwd(const float* p, const float* target, long long n, float coef, const float* gscale, float* d_p, void* stream);
/* F.nll_loss on log-probs [rows][classes] (multimnist/train.py:79) */
int mmvae_nll_fwd(const float* logp, const long long* target, int rows, int classes, float* out_sum, void* stream);
int mmvae_nll_bwd(const long long* target, int rows, int classes, float coef, const float* gscale, float* d_logp, void* stream);
/* counter-based RNG (Philox4x32-10) */
int mmvae_normal(float* out, long long n, unsigned long long seed, const long long* step_counter, unsigned stream_id, void* stream);
int mmvae_keep_mask(uint8_t* out, long long n, float p, unsigned long long seed, const long long* step_counter,
                    unsigned stream_id, void* stream);
/* F.mse_loss of the COCO loss variant (coco/train.py:75): out[0] += sum (a-b)^2 ; bwd: d_a = coef * 2 (a-b) */
int mmvae_mse_fwd(const float* a, const float* b, long long n, float* out_sum, void* stream);
int mmvae_mse_bwd(const float* a, const float* b, long long n, float coef, const float* gscale, float* d_a, void* stream);
/* Input pipeline: the ToTensor() transform of the reference's loaders (multimnist/train.py:113-121; dataset tensors are
 * uint8 (N,50,50), multimnist/datasets.py:180-181) done on the device: dst[i] = src[i] / denom (denom = 255, IEEE division: bit-equal to ToTensor) */
int mmvae_u8_to_f32(const uint8_t* src, long long n, float denom, float* dst, void* stream);
/* ... behind `wait_event` (a hipEvent_t, or NULL): `stream` waits for the event, then converts -- one call of the enqueue thread per batch */
/* Batch gather + ToTensor in one kernel: dst[r][i] = src[idx[r]][i] / denom for `rows` rows of `row_elems` uint8 elements (a multiple of 4).
 * `src` and `idx` may be PINNED HOST memory: the device reads the rows over the host link itself -- no runtime copy call, no
 * staging buffer, no conversion kernel on the compute stream (the loader's device-gather path for small batches). */
int mmvae_gather_rows_u8_f32(const uint8_t* src, const long long* idx, long long rows, long long row_elems, float denom, float* dst, void* stream);
int mmvae_u8_to_f32_after(const uint8_t* src, long long n, float denom, float* dst, void* wait_event, void* stream);
/* One staged batch to the device: up to two asynchronous host-to-device copies (pinned sources; bytes_b may be 0) on `stream`, then
 * `event` (a hipEvent_t) recorded behind them.  One call of the loader's worker thread per batch (coco/train.py:117-128,144-147: the
 * reference's DataLoader + .cuda() per batch): a foreign-function call holds no interpreter lock while the runtime takes its own. */
int mmvae_h2d_stage(void* dst_a, const void* src_a, size_t bytes_a, void* dst_b, const void* src_b, size_t bytes_b, void* event, void* stream);
/* Events for a host-side pipeline next to the step (the loader's "copy done" / "buffers consumed" edges): created WITHOUT the
 * system-scope fence a default event performs when it completes (a write-back of the device caches for the host's benefit: one per
 * step on the compute stream slowed the kernels behind it by tens of microseconds) and without timing.  They order streams of one
 * device and tell the host THAT work finished, not what it wrote -- read results after a stream synchronisation. */
int mmvae_stream_wait_event(void* stream, void* event);   /* device-side: `stream` waits for `event` */
int mmvae_event_create(void** out);
int mmvae_event_destroy(void* event);
int mmvae_event_record(void* event, void* stream);
int mmvae_event_synchronize(void* event);            /* blocks the calling host thread */
/* torch.optim.Adam defaults (multimnist/train.py:129,173) on flat buffers; state = device int64[2] {step, ticket},
 * zero-initialised by the caller; grad_scale multiplies g first (1/world_size after a sum all-reduce). */
/* The three losses of train()'s closure from a step's `sums` block (multimnist/train.py:33-62: the weighted sum each
 * loss_function call ends with): losses[k] = w_bce[k]*sums[k] + w_nll[k]*sums[4+k] + w_kl[k]*sums[8+k].  The weights are HOST
 * arrays of 3 floats (lambda / divisor per pass, 0 for an absent pass); `sums` and `losses` are device pointers. */
int mmvae_step_losses(const float* sums, const float* w_bce, const float* w_nll, const float* w_kl, float* losses, void* stream);
/* Measurement aids (not part of the reference's interface).  mmvae_debug_set: named integer A/B switches of the library
 * (DESIGN.md lists them).  mmvae_debug_probe(1): every kernel the library launches through its tagged launcher carries a start
 * and a stop event of its own until mmvae_debug_probe(0); mmvae_debug_probe_read waits for them and writes one line per launch
 * "tag\tkernel\tmicroseconds\talgorithmic FLOPs\n" into buf (returns the number of lines) -- bench.py's in-step roofline. */
/* ONE more HIP stream (non-blocking, default priority) for host code that needs a copy / communication stream next to the
 * step's: a PyTorch host must not take it from torch's pool -- the first torch.cuda.Stream() creates the pool's 32 streams, past
 * the hardware queues the runtime maps streams to (GPU_MAX_HW_QUEUES) every stream of the process is then time-sliced and each
 * step runs ~2x slower (measured 0.67 -> 1.26 ms, tools/loader_probe.py).  Wrap it with torch.cuda.ExternalStream. */
int mmvae_stream_create(void** out);
int mmvae_stream_destroy(void* stream);
/* Batch assembly for the input pipeline (coco/train.py:117-128's DataLoader role): dst[i] = src[idx[i]], rows of row_bytes (a
 * multiple of 16).  `idx` and `dst` are device memory; `src` may be PINNED HOST memory -- the kernel then pulls the rows over the
 * host link itself, and the host never copies a sample (data.DeviceBatcher). */
int mmvae_gather_rows(const void* src, const long long* idx, long long rows, long long row_bytes, void* dst, void* stream);
/* Waits for `stream`, then MMVAE_OK, or MMVAE_ETIMEOUT when the step that wrote `sums` (device, [16]) declared itself void. */
int mmvae_step_status(const float* sums, void* stream);
int mmvae_debug_probe(int on);
int mmvae_debug_probe_read(char* buf, long long cap);
/* torch.optim.Adam (multimnist/train.py:129,173) on flat fp32 buffers.  `state`: 16 zero-initialised device bytes the caller keeps
 * next to the moments -- int64 step count, then two words of the kernel's own (block ticket, skip flag).  The update is dropped
 * (and the step count kept) when the skip flag is set or g[0] carries the void mark (the NaN bit pattern 0x7fc0dead, sign ignored):
 * see mmvae_coco_step_io.optimizer_state.  Any other NaN in the gradient is a value: it goes through the update and shows in the
 * parameters, as with torch.optim.Adam. */
int mmvae_adam_step(float* p, const float* g, float* m, float* v, long long n, long long* state, float lr, float beta1,
                    float beta2, float eps, float grad_scale, void* stream);
/* The same update with the unpack of the packed weight gradients fused in (mmvae_mm_step_io.defer_unpack = 1):
 * g[i] += packed gradient of i (through gmap), the completed g is written back, then Adam.  One pass instead of two. */
int mmvae_adam_step_packed(float* p, float* g, float* m, float* v, long long n, long long* state, float lr, float beta1,
                           float beta2, float eps, float grad_scale, const int* gmap, const float* gpk, const float* gpk_vec,
                           void* stream);
/* ... over `nr` (<= 4) element ranges (offset, length: multiples of 4 elements) of the flat buffers of n elements.  advance = 1: this call
 * completes the optimizer step (the step count advances); 0: an earlier part of it. */
int mmvae_adam_step_packed_ranges(float* p, float* g, float* m, float* v, long long n, const long long* ranges, int nr, int advance,
                                  long long* state, float lr, float beta1, float beta2, float eps, float grad_scale, const int* gmap,
                                  const float* gpk, const float* gpk_vec, void* stream);
/* The (offset, length) runs of the flat buffers that mmvae_mm_step_io.early_adam updates inside the step (image_decoder.*, text_decoder.*):
 * writes up to `cap` pairs to `ranges`, returns their number. */
int mmvae_mm_early_ranges(const mmvae_mm_t*, long long* ranges, int cap);

#ifdef __cplusplus
}
#endif
#endif
