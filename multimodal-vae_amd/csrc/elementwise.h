// Streaming (HBM/L2-bound) kernels around the MFMA GEMMs: thin-layer im2col, BatchNorm finalize / backward
// apply, sigmoid+BCE, product-of-experts / reparametrisation / KL, Adam, weight packing.  See elementwise.hip.
#pragma once
#include "common.h"

// ---- weight packing (fp32 reference-layout master  <->  bf16 GEMM-layout copies / fp32 packed grads) ----
struct PackDesc {
    long long dst_off;      // element offset in the packed buffer (bf16 units for matrices, float units for vectors)
    long long src_off;      // element offset in the flat fp32 parameter buffer
    int Npad, Kpad, N, K;   // packed matrix [Npad][Kpad]; entries outside [N][K] are zero
    int NL;                 // row n = n_hi*NL + n_lo
    int s_nhi, s_nlo;       // source strides of n_hi / n_lo
    int TW, C;              // column k = (ty*TW + tx)*C + c
    int s_ty, s_tx, s_c;    // source strides
    int o_ty, o_tx, step_t; // kernel index = o + t*step
    int is_f32;             // destination is fp32 (bias vectors, Kpad == 1)
    long long bias_off;     // >= 0: column K of every row holds a bias folded into the GEMM (operand column K == 1.0)
    int b_nhi, b_nlo;       // source strides of that bias vector
    int first_block;        // prefix sum of 256-thread blocks over the table
    int part;               // gradient descriptors: 1 = complete early in the step (decoders), 0 = with the encoders' backward;
                            // weight descriptors: the stage of the step that refreshes the copy (StepBeginArgs::pack_parts), 0 = the prologue
    int frag;               // 1: MFMA-fragment-major destination -- 16-row tile nt, 32-column k-step ks: the 64 lanes' 8-element
                            // vectors (lane = 16*(k/8 % 4) + n % 16) are one contiguous 1 KB block at ((nt*(Kpad/32) + ks)*64 + lane)*8
};

int launch_pack(const PackDesc* table_dev, const PackDesc* table_host, int ndesc, const float* params,
                bf16* packed_bf, float* packed_f32, hipStream_t s);
// the descriptors [d0, d1) of the table only (a plan whose streams need different parts of the packed weights first)
int launch_pack_range(const PackDesc* table_dev, const PackDesc* table_host, int ndesc, int d0, int d1, const float* params,
                      bf16* packed_bf, float* packed_f32, hipStream_t s);
// grads_flat[src] += packed_grad[dst]  (inverse of the matrix packing, fp32 -> fp32)
// part >= 0: only the descriptors with PackDesc::part == part (the data-parallel step hands the early part to the collective
// while the rest of the backward pass runs)
int launch_unpack_grads(const PackDesc* table_dev, const PackDesc* table_host, int ndesc, const float* gpk_mat,
                        const float* gpk_vec, float* grads_flat, hipStream_t s, int part = -1);

int launch_im2col_small(const float* src, int Nimg, int Cin, int H, int W, int KH, int KW, int stride, int pad,
                        int OH, int OW, bf16* dst, int ld, hipStream_t s);

struct BnFinalizeArgs {
    const float2* stats;     // [G][C] (sum, sumsq) of the raw conv/linear output
    int G, C;
    float count;             // elements per channel per group
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; long long* num_batches_tracked;   // may be null (no update)
    int updates_per_group;   // how many forward calls each group's statistics stand for (encoder dedup: 2)
    unsigned skip_update_mask;   // bit g set: group g's forward does not exist in the reference step (weak-supervision pass masks):
                             // its statistics normalise the (unused) rows but do not touch the running buffers
    float2* affine;          // [G][C] (scale, shift):  y = x*scale + shift
    float2* meanrstd;        // [G][C]
    float eps, momentum;
    int training;            // 0: affine from running statistics (stats ignored)
};
int launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s);

// Fused BatchNorm finalize + normalise + activation: a = act(gamma*(r-mean)*rstd+beta) materialised ONCE per layer
// (the GEMM operand staging is then a pure copy).  Also emits the per-(group,channel) tables backward needs and
// updates the running statistics.  In eval mode the statistics come from the running buffers.
struct BnActArgs {
    const bf16* r;           // [rows][ld] raw conv/linear output
    bf16* a;                 // [rows][ld] activated output
    int rows, C, ld, rows_per_group, G;
    int act;
    BnFinalizeArgs fin;      // stats, gamma/beta, running buffers, tables (count = elements per channel per group)
};
int launch_bn_act(const BnActArgs& a, hipStream_t s);

struct BnBwdApplyArgs {
    const bf16* db;          // [rows][ld] grad wrt BN output (after d-activation)
    const bf16* db2;         // optional second contribution for the same rows (encoder features shared by 2 passes)
    const bf16* r;           // [rows][ld] raw (pre-BN) tensor
    bf16* dr;                // [rows][ld] out (may alias db)
    int rows, C, ld, rows_per_group, G;
    const float2* red;       // [G][C] (sum db, sum db*xhat)
    const float2* meanrstd;  // [G][C]
    const float* gamma;
    float* dgamma; float* dbeta;   // flat grads, += (may be null)
    float* dbias;            // optional: += column sums of dr (Linear bias in front of the BN), may be null
};
int launch_bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t s);

struct BceArgs {
    const float* logits;     // NHWC [G*B][H][W][ldl] (channel c at +c)
    int ldl;
    const float* target;     // NCHW [B][C][H][W], shared by all groups
    int G, B, C, H, W;
    float* recon;            // NCHW [G*B][C][H][W] or null
    float* dlogit;           // NCHW [G*B][C][H][W] or null:  coef[g] * dBCE/dlogit
    float coef[4];
    float* loss_sum;         // [G] += sum of BCE terms
};
int launch_sigmoid_bce(const BceArgs& a, hipStream_t s);

// ---- drop-in granular latent ops (ProductOfExperts / reparametrize / KL) ----
int launch_poe_fwd(const float* mu, const float* logvar, int M, int n, float* out_mu, float* out_logvar, hipStream_t s);
int launch_poe_bwd(const float* mu, const float* logvar, int M, int n, const float* g_mu, const float* g_logvar,
                   float* d_mu, float* d_logvar, hipStream_t s);
int launch_reparam_fwd(const float* mu, const float* logvar, const float* eps, int n, float* z, hipStream_t s);
int launch_reparam_bwd(const float* logvar, const float* eps, const float* dz, int n, float* d_mu, float* d_logvar, hipStream_t s);
int launch_kl_fwd(const float* mu, const float* logvar, int n, float* kl_sum, hipStream_t s);
int launch_kl_bwd(const float* mu, const float* logvar, int n, float coef, const float* gscale, float* d_mu, float* d_logvar, hipStream_t s);
int launch_normal(float* out, long long n, unsigned long long seed, const long long* step_counter, unsigned stream_id, hipStream_t s);
int launch_keep_mask(uint8_t* out, long long n, float p, unsigned long long seed, const long long* step_counter,
                     unsigned stream_id, hipStream_t s);

// ---- fused 3-pass latent block of the ELBO step (multimnist/train.py:154-166) ----
struct Latent3Args {
    int B, D;
    const float* img_out;    // [2][B][2D]  image encoder (mu|logvar) of pass 1 and pass 2
    const float* img_out_b;  // optional: pass-2 encoder output elsewhere (null -> img_out + B*2D)
    const float* txt_out;    // [B][2D]
    const float* eps;        // [3][B][D]
    float* mu; float* logvar;   // [3][B][D]
    float* z_f32;            // [3B][D]
    bf16* z_bf;              // [3B][ldz], pad columns zeroed
    int ldz;
    float* kl_sum;           // [3] +=
    int training;
};
int launch_latent3_fwd(const Latent3Args& a, hipStream_t s);
struct Latent3BwdArgs {
    Latent3Args f;
    const float* dz_a;       // [3B][D] (image decoder) may be null
    const float* dz_b;       // [3B][D] (text decoder) may be null
    float kl_coef[3];        // kl_lambda / B (or the MNIST divisor) per pass
    bf16* d_img_out_bf;      // [2][B][2D]
    float* d_img_bias;       // [2D] += column sums over both passes (classifier last bias), may be null
    float* d_txt_out;        // [B][2D] or null
    int sum_img_variants;    // 1: d_img_out_bf is [B][2D] = pass-1 + pass-2 gradient
    bf16* d_txt_out_bf;      // optional bf16 copy [B][2D]
    float* d_txt_bias;       // optional [2D] += column sums of the text-encoder output gradient
    float* d_img_out_f32;    // optional fp32 [B][2D] = pass-1 + pass-2 gradient (then d_img_out_bf is not written)
    const float* loss_slots; // optional: the step's [MMVAE_LOSS_SLOTS][16] loss accumulators, summed into loss_out[16] by block 0
    float* loss_out;         // (every loss term must be final when this kernel starts: saves the separate sum_slots launch)
};
int launch_latent3_bwd(const Latent3BwdArgs& a, hipStream_t s);

// bit pattern of the "void step" mark in element 0 of a flat gradient (plan_base.h sum_slots_kernel): a quiet NaN with a payload no
// arithmetic produces on its own
#define MMVAE_VOID_MARK 0x7fc0deadu
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long long n;
    long long* step;         // device counter, incremented by the kernel (first call sees 0 -> t=1)
    float lr, b1, b2, eps;
    float grad_scale;        // applied to g before the update (1/world for data-parallel averaging)
    // optional: the part of the gradient that is still in the packed weight-gradient buffers (fused unpack):
    // g_total[i] = g[i] + gpk[gmap[i]] (gmap >= 0) or + gpk_vec[-gmap[i]-2] (gmap < -1); written back to g_out[i]
    const int* gmap; const float* gpk; const float* gpk_vec; float* g_out;
    // optional: the update covers `nr` element ranges (offset, length: multiples of 4) of the flat buffers instead of [0, n) -- an
    // optimizer step issued in parts (the decoders' parameters while the encoders' backward still runs).  Only the part with
    // `advance` set counts the step; 0 ranges = the whole buffer, advancing.
    int nr; long long roff[4], rlen[4]; int no_advance;
};
int launch_adam(const AdamArgs& a, hipStream_t s);
// dst[i] = src[idx[i]] for `rows` rows of row_bytes (a multiple of 16) each; src may be pinned HOST memory (the device reads it
// over the host link: the loader's batch gather without a host-side copy)
int launch_gather_rows(const void* src, const long long* idx, long long rows, long long row_bytes, void* dst, hipStream_t s);
// dst[r][i] = src[idx[r]][i] / denom (uint8 rows of a multiple of 4 elements -> fp32 rows): gather and ToTensor in one kernel
int launch_gather_rows_u8_f32(const unsigned char* src, const long long* idx, long long rows, long long row_elems, float denom, float* dst, hipStream_t s);
// loss_k = w_bce[k] * sums[k] + w_nll[k] * sums[4+k] + w_kl[k] * sums[8+k]   (the closing arithmetic of loss_function, train.py:33-62)
struct StepLossArgs { const float* sums; float* out; float w_bce[3], w_nll[3], w_kl[3]; };
int launch_step_losses(const StepLossArgs& a, hipStream_t s);
// map[flat parameter index] = location of its packed gradient (see AdamArgs), -1 where there is none
int launch_unpack_map(const PackDesc* table_dev, const PackDesc* table_host, int nd, long long nparams, long long gmat_elems, int* map, hipStream_t s);
int launch_fill_zero(void* p, size_t bytes, hipStream_t s);
int launch_bce_fwd(const float* p, const float* t, long long n, float* out, hipStream_t s);
int launch_bce_bwd(const float* p, const float* t, long long n, float coef, const float* gscale, float* dp, hipStream_t s);
int launch_nll_fwd(const float* lp, const long long* tg, int rows, int classes, float* out, hipStream_t s);
int launch_nll_bwd(const long long* tg, int rows, int classes, float coef, const float* gscale, float* dlp, hipStream_t s);

// One launch at the top of the fused step: zero the accumulation buffers (16-byte stores) and draw the step's
// stochastic inputs (eps ~ N(0,1), dropout keep flags) from Philox streams keyed by (seed, step counter).
struct StepBeginArgs {
    void* zero_ptr[4]; size_t zero_bytes[4];      // multiples of 16 bytes, 16-byte aligned (unused: null/0)
    float* eps; long long n_eps;                  // null -> not drawn
    uint8_t* mask[3]; long long n_mask[3];        // null -> not drawn
    float p;
    unsigned long long seed; const long long* step;
    // optional: the weight pack of the previous optimizer step rides in the same launch (step_begin_with_pack)
    const PackDesc* pack_table; int pack_nd; const float* pack_params; bf16* packed_bf; float* packed_f32; int pack_blocks;
    unsigned pack_parts;                          // 0: every descriptor; else bit p: the descriptors with PackDesc::part == p (a step that
                                                  // refreshes the copies its first kernels read in the prologue and the rest on a side stream)
};
int launch_step_begin(const StepBeginArgs& a, hipStream_t s);
int step_begin_with_pack(StepBeginArgs& a, const PackDesc* table_dev, const PackDesc* table_host, int nd, const float* params,
                         bf16* packed_bf, float* packed_f32);

// ---- small ops of the MLP models (mnist/model.py:136-170) ----
// x[r][:] = table[idx[r % idx_rows]][:] as bf16 rows of stride ld (pad columns zero) + BatchNorm column statistics
int launch_embed_gather_stats(const float* table, int C, const long long* idx, int rows, int idx_rows, int rows_per_group,
                              bf16* x, int ld, float2* colstats, hipStream_t s);
// g_table[idx[r]][c] += d[r][c]
int launch_embed_scatter_add(const bf16* d, int ld, int C, const long long* idx, int rows, float* g_table, hipStream_t s);
// log_softmax over `classes` columns of logits [rows][classes] (+ NLL against target[r % target_rows]):
// words = log-probs, nll_sum[g] += sum(-logp[target]) per group, dlogits (bf16 [rows][ld_d], pads zero) = coef[g]*(softmax - onehot)
struct LogSoftmaxNllArgs {
    const float* logits; int rows, classes;
    float* words;                      // [rows][classes]
    const long long* target; int target_rows; int rows_per_group;
    float* nll_sum;                    // [MMVAE_LOSS_SLOTS][16] column g, or null
    bf16* dlogits; int ld_d;           // or null
    float* dlogits_f32;                // optional fp32 copy [rows][classes]
    float coef[4];
};
int launch_logsoftmax_nll(const LogSoftmaxNllArgs& a, hipStream_t s);
int launch_cast_bf16(const float* x, long long n, bf16* out, hipStream_t s);
// out[c] += sum_r x[r][c]   (x: fp32 [rows][cols]) -- bias gradients of the last Linear of a stack
int launch_colsum_f32(const float* x, int rows, int cols, float* out, hipStream_t s);
int launch_colsum_bf16(const bf16* x, int ld, int rows, int cols, float* out, hipStream_t s);
// input pipeline (multimnist/datasets.py:45-74 + torchvision ToTensor): uint8 pixels -> fp32 / denom
int launch_u8_to_f32(const uint8_t* src, long long n, float denom, float* dst, hipStream_t s);
// F.mse_loss pieces of the COCO loss variant (coco/train.py:75)
int launch_mse_fwd(const float* a, const float* b, long long n, float* out_sum, hipStream_t s);
int launch_mse_bwd(const float* a, const float* b, long long n, float coef, const float* gscale, float* d_a, hipStream_t s);
