// Parameter blocks of the two MFMA workhorse kernels (gemm.hip):
//   gemm_gather : C[rows][N] = act(A_gathered)[rows][K] * Wp[N][K]^T   (conv fwd / conv dgrad / convT / Linear)
//   wgrad       : dWp[N][K] += P[rows][N]^T * act(G_gathered)[rows][K]  (all weight gradients)
// "rows" are pixels (n, oy, ox) of a row grid; the gathered operand is an NHWC bf16 tensor whose pixel for
// row (n,oy,ox) and tap (ty,tx) is (n, oy*sy+offy+ty*dy, ox*sx+offx+tx*dx); K = TH*TW*C with c fastest.
#pragma once
#include "common.h"
#include "elementwise.h"
#include <vector>

#define MMVAE_MAX_CLASSES 4

struct GatherClass {        // one stride-parity class (conv fwd and Linear have exactly one)
    int OY, OX;             // row grid per image
    int rows_per_group;     // group_n*OY*OX
    int TH, TW;             // taps
    int offy, offx;         // gather offsets
    int ooy, oox;           // position of row (oy,ox) in the "row tensor": (oy*osy+ooy, ox*osx+oox)
    int K;                  // TH*TW*C
    int Kpad;               // packed weight row stride (multiple of 64)
    const bf16* Wp;         // gemm: packed weights [Npad][Kpad]
    const bf16* Wf;         // gemm: optional fragment-major copy of Wp (PackDesc::frag = 1) for the direct-B convres kernels
    float* dWp;             // wgrad: packed fp32 gradient [Npad][Kpad]
};

struct GatherCommon {
    int groups, group_n;    // BN groups (passes) and images per group
    // gathered tensor [Nimg][AH][AW][Ald] bf16
    const bf16* A;
    int AH, AW, Ald, C;
    int sy, sx, dy, dx;
    const float2* a_affine; // [groups][C] (scale, shift) applied before a_act; may be null
    int a_act;
    const uint8_t* a_mask;  // keep flags [rows][C] (single-tap dense operands only); may be null
    float a_mask_scale;
    int a_bcast_n;          // >0: the gathered tensor holds only a_bcast_n images shared by every row block
                            // (image = row_image % a_bcast_n, affine group 0): encoder features reused by 2 passes
    // row tensor geometry [Nimg][OH][OW][ld]: gemm output / wgrad plain operand
    int OH, OW, osy, osx;
    int N;                  // valid columns
    int nclasses;
};

// Transform of the gathered tensor applied while it is staged (image-resident conv kernels only, convres.hip): the
// elementwise kernel that used to materialise the operand leaves the chain.
struct GatherTransform {
    int kind;                   // 0 none
                                // 1: the tensor is a RAW conv output; staged value = Swish(BatchNorm(x)), tables from `fin` (the launch
                                //    also writes fin.affine / fin.meanrstd and updates the running statistics, like bn_act_kernel)
                                // 2: the tensor is db (gradient w.r.t. a BatchNorm output); staged value = the BatchNorm-backward dr
    BnFinalizeArgs fin;         // kind 1
    const bf16* r;              // kind 2: raw tensor the BatchNorm normalised (geometry of the gathered tensor)
    const float2* red;          //         [groups][SLOTS][C] (sum db, sum db*xhat)
    const float2* mr;           //         [groups][C] (mean, rstd)
    const float* gamma;
    float* dgamma; float* dbeta;    //     += parameter gradients (may be null)
    float inv_cnt;              //         1 / elements per channel per group
    int groups;
    bf16* out;                  // kind 1 / 2, optional: the staged (transformed) tensor is also written here, same layout as the gathered
                                // tensor -- the operand the layer's weight gradient reads, as a by-product instead of a kernel of its own
};

struct GemmParams {
    GatherCommon c;
    GatherClass cls[MMVAE_MAX_CLASSES];
    // split-K for small row counts (few workgroups, long K): `ksplit` workgroups share one output tile; partial
    // tiles are added into zeroed fp32 scratch with float atomics, an arrival ticket elects the last workgroup,
    // which runs the epilogue and leaves scratch + ticket zeroed again.  sk_buf: tiles*128*BN floats.
    int ksplit;
    float* sk_buf;
    unsigned* sk_cnt;
    const float* bias;      // [N] or null
    bf16* out_bf;           // [Nimg][OH][OW][ldo] or null
    float* out_f;           // same geometry, fp32, or null
    int ldo;
    bf16* out_act_bf;       // optional second output act(v) [* keep*scale]: the next layer's operand (no BN in between)
    int e_act;
    const uint8_t* e_mask;  // [rows][N] keep flags or null
    float e_mask_scale;
    float2* colstats;       // [groups][MMVAE_STAT_SLOTS][N] += (sum v, sum v^2) or null
    // d-activation epilogue: v *= act'(affine(r)) [* keep*scale], r has the output geometry (ld = d_ld)
    const bf16* d_r;
    int d_ld;
    int d_bcast_n;          // >0: d_r holds d_bcast_n images (image = row_image % d_bcast_n)
    const float2* d_affine; // [groups][N] or null
    int d_act;
    const uint8_t* d_mask;  // [rows][N] keep flags or null
    float d_mask_scale;
    const float2* d_meanrstd;  // [groups][N] for xhat=(r-mean)*rstd, with d_red
    float2* d_red;             // [groups][MMVAE_STAT_SLOTS][N] += (sum v, sum v*xhat) or null
    float* d_colsum;           // [N] += sum v over all rows (bias gradient of the producer Linear) or null
    int npad;                  // rows of the packed weight matrices (gemm_small range-checks weight rows against it)
    const GatherTransform* tr; // host-side only (read by the launcher): staging transform of the gathered tensor, or null
    int d_cmod;                // >0: the BatchNorm tables (d_affine/d_meanrstd/d_red) have d_cmod channels and output
                               // column n belongs to channel n % d_cmod (Linear over a flattened NHWC feature map)
};

struct WgradParams {
    GatherCommon c;
    GatherClass cls[MMVAE_MAX_CLASSES];
    const bf16* P;          // plain operand, row tensor geometry, ld = ldp
    int ldp;
    const float2* p_affine; // optional transform of the plain operand [groups][N]
    int p_act;
    int rows_per_block;     // multiple of 64
    // slab form (filled by the launcher): partial tiles go to slab + chunk*slab_chunk_stride + slab_cls_off[class] as
    // [rows][Kpad] with plain stores and wgrad_reduce_kernel sums the chunk copies into dWp; null: fp32 atomics into dWp
    float* slab;
    long long slab_chunk_stride;
    long long slab_cls_off[MMVAE_MAX_CLASSES];
};

// Slab pool of one step (carved from the caller's workspace) and the reductions its launches owe
struct WgradSlabJob { float* dst; const float* slab; int N, K, Kpad, chunks; long long chunk_stride; hipStream_t stream;
                      int src_ld; };        // row length of a slab copy (0: Kpad)
// ... and of the ring-staged kernel (wgrad_ring.hip): one job per stride-parity class, `copies` compact [N][Kc] partial copies whose
// column k*C + c belongs to tap (ty0 + (k / ntx)*st, tx0 + (k % ntx)*st) of the layer
struct WgradRingJob { float* dst; const float* slab; int N, Kpad, C, Kc, copies, ntx, ty0, tx0, st, kw; hipStream_t stream; };
struct WgradSlabCtx {
    float* pool = nullptr;
    size_t cap = 0, used = 0;           // in floats
    std::vector<WgradSlabJob> jobs;
    std::vector<WgradRingJob> ring_jobs;
    void reset(float* p, size_t c) { pool = p; cap = c; used = 0; jobs.clear(); ring_jobs.clear(); }
    float* take(size_t n) {             // n floats of the pool (16-byte granules) or null when it is exhausted
        n = (n + 3) / 4 * 4;
        if (!pool || used + n > cap) return nullptr;
        float* p = pool + used;
        used += n;
        return p;
    }
};

int launch_gemm_gather(const GemmParams& p, hipStream_t stream);
// gemm_small.hip: problems too small to fill the chip with 128-row tiles (classifier / bottleneck layers): one output
// tile per wave, both operands loaded straight into MFMA fragments, K optionally split over the waves of a workgroup
int try_launch_gemm_small(const GemmParams& p, hipStream_t stream);
int launch_wgrad(const WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx = nullptr);
// n independent problems; the ones of the 128x128 tile class share ONE launch
int launch_wgrad_group(const WgradParams* list, int n, hipStream_t stream, WgradSlabCtx* ctx = nullptr);
// sums the slab copies into the packed gradients, one launch.  only_own = true: just the slabs whose kernels were issued
// on `stream` itself (safe right behind them); false: every slab still owed -- the caller has joined the streams
int launch_wgrad_reduce(WgradSlabCtx* ctx, hipStream_t stream, bool only_own = false);
