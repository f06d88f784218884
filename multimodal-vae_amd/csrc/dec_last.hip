// Fused tail of the MultiMNIST image decoder on the matrix cores (thin.h: DecLastFusedArgs, 25x25x32 -> 50x50x1).
//
// dec_last_fused_kernel (thin.hip) computes the thin ConvTranspose2d(32, 1, 4, 2, 1), its input gradient and its weight
// gradient with vector dot products per output pixel: 66-75 us per step, the longest kernel of the main chain, VALU- and
// latency-bound at 4 waves per SIMD.  All three products are tiny GEMMs once the 16 taps are made the narrow dimension:
//   forward      P[pixel][tap]   = A[pixel][32 ch] . W[32 ch][16 taps]          one 32x32x16 MFMA pair per 32 pixels
//                logit[oy][ox]   = the 4 entries of P whose (pixel, tap) land on (oy, ox)        (overlap-add out of LDS)
//   input grad   dA[pixel][ch]   = patch[pixel][16 taps] . W^T[16 taps][32 ch]  one MFMA per 32 pixels, patch = the 4x4
//                                  window of dlogit around the pixel
//   weight grad  dW[ch][tap]     = sum over pixels A[pixel][ch] . patch[pixel][tap]              transposed LDS reads
// One workgroup (8 waves) owns one image: raw input -> BatchNorm + Swish while staging (A is never written to memory),
// P and the patches live in LDS, the input gradient leaves through the accumulator epilogue of the conv kernels
// (d-Swish of the producer, BatchNorm-backward sums, 16-byte stores), the weight-gradient partial goes to the slab.
#include "thin.h"
#include "bn_dev.h"
#include "convres_epi.h"

namespace {

constexpr int DL_IH = 25, DL_IW = 25, DL_OH = 50, DL_OW = 50, DL_C = 32, DL_NPIX = 625, DL_TILES = 20, DL_ROWS = 640;
constexpr int DL_AP = 80;              // A tile: bytes per pixel (32 bf16 + 16)
constexpr int DL_PP = 48;              // patches: bytes per pixel (16 bf16 + 16)
constexpr int DL_DLW = 52;             // dlogit rows with a zero halo
constexpr int DL_WAVES = 8, DL_NTHR = DL_WAVES * 64;
// LDS map
constexpr int DL_OFF_A = 0;                                        // [640][80]
constexpr int DL_OFF_P = DL_OFF_A + DL_ROWS * DL_AP;               // P fp32 [640][16]; later the patches [640][48]
constexpr int DL_OFF_DL = DL_OFF_P + DL_ROWS * 64;                 // dlogit fp32 [52][52]
constexpr int DL_OFF_SCR = DL_OFF_DL + DL_DLW * DL_DLW * 4;        // per-wave epilogue scratch [8][2560]; later dW partials [8][32][16] fp32
constexpr int DL_OFF_TAB = DL_OFF_SCR + DL_WAVES * 2560;           // BatchNorm tables float2 [32] x 2
constexpr int DL_LDS = DL_OFF_TAB + 2 * DL_C * 8;
static_assert(DL_ROWS * DL_PP <= DL_ROWS * 64, "the patches reuse the P buffer");
static_assert(DL_LDS <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) s16x4 lds_s16x4d;
__device__ __forceinline__ bf16x8 tr_pair_d(const char* a0, const char* a1) {
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4d*)a0);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4d*)a1);
    return u.v;
}

__global__ __launch_bounds__(DL_NTHR) void dec_last_mfma_kernel(const DecLastFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const A_s = smem + DL_OFF_A;
    float* const P_s = reinterpret_cast<float*>(smem + DL_OFF_P);
    char* const pat_s = smem + DL_OFF_P;
    float* const dl_s = reinterpret_cast<float*>(smem + DL_OFF_DL);
    char* const scr_s = smem + DL_OFF_SCR;
    float2* const aff_s = reinterpret_cast<float2*>(smem + DL_OFF_TAB);
    float2* const mr_s = aff_s + DL_C;
    __shared__ float part[DL_WAVES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n_in_g = blockIdx.x, g = blockIdx.y;
    const long long n = (long long)g * a.B + n_in_g;
    const BnFinalizeArgs& f = a.fin;
    const bool bwd = g < a.bwd_groups;

    // ---- raw image: loads in flight while the tables are made and the buffers cleared
    constexpr int NV = DL_NPIX * 4, IT = (NV + DL_NTHR - 1) / DL_NTHR;          // 16-byte vectors of the image
    i32x4c rv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int v = tid + it * DL_NTHR;
        if (v < NV) rv[it] = *reinterpret_cast<const i32x4c*>(a.r + (size_t)n * DL_NPIX * DL_C + (size_t)v * 8);
    }
    if (tid < DL_C) {
        float2 aff, mr;
        bn_channel_tables(f, g, tid, aff, mr);
        aff_s[tid] = aff; mr_s[tid] = mr;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {                // the tables backward reads + running statistics: once
        for (int i = tid; i < f.G * DL_C; i += DL_NTHR) {
            float2 aff, mr;
            bn_channel_tables(f, i / DL_C, i % DL_C, aff, mr);
            f.affine[i] = aff; f.meanrstd[i] = mr;
        }
        bn_running_update(f, tid, DL_NTHR);
    }
    {
        const i32x4c z = {0, 0, 0, 0};
        for (int i = tid * 16; i < (DL_ROWS - DL_NPIX) * DL_AP; i += DL_NTHR * 16)      // padding pixel rows of A
            *reinterpret_cast<i32x4c*>(A_s + DL_NPIX * DL_AP + i) = z;
        for (int i = tid; i < DL_DLW * DL_DLW; i += DL_NTHR) dl_s[i] = 0.f;
    }
    // weight fragments (fp32 (32, 1, 4, 4) -> bf16): forward B[k = ch][j = tap] for the two k-steps, input gradient B[k = tap][j = ch]
    bf16x8 wf[2], wd;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[ks][j] = r < 16 ? (bf16)a.w[(ks * 16 + 8 * h + j) * 16 + r] : (bf16)0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) wd[j] = (bf16)a.w[r * 16 + 8 * h + j];
    __syncthreads();
    // ---- stage: BatchNorm + Swish -> A tile
    {
        const int cv = tid & 3;                              // DL_NTHR % 4 == 0: a thread keeps its channel octet
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float2 t = aff_s[cv * 8 + j]; sc[j] = t.x; sh[j] = t.y; }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = tid + it * DL_NTHR;
            if (v < NV) {
                const bf16x8 x = __builtin_bit_cast(bf16x8, rv[it]);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)swish_fast((float)x[j] * sc[j] + sh[j]);
                *reinterpret_cast<bf16x8*>(A_s + (v >> 2) * DL_AP + cv * 16) = o;
            }
        }
    }
    __syncthreads();
    // ---- forward: P[pixel][tap]
    for (int t = wave; t < DL_TILES; t += DL_WAVES) {
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const char* ap = A_s + (t * 32 + r) * DL_AP + h * 16;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap), a1 = *reinterpret_cast<const bf16x8*>(ap + 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[1], acc, 0, 0, 0);
        if (r < 16) {                                        // lane = tap, register j = pixel (j&3) + 8*(j>>2) + 4h of the tile
#pragma unroll
            for (int j = 0; j < 16; ++j) P_s[(t * 32 + (j & 3) + 8 * (j >> 2) + 4 * h) * 16 + r] = acc[j];
        }
    }
    __syncthreads();
    // ---- logits by overlap-add, sigmoid, BCE and its gradient
    float loss = 0.f;
    for (int o = tid; o < DL_OH * DL_OW; o += DL_NTHR) {
        const int oy = o / DL_OW, ox = o - oy * DL_OW;
        const int kh0 = (oy + 1) & 1, kw0 = (ox + 1) & 1;
        const int iy0 = (oy + 1 - kh0) >> 1, ix0 = (ox + 1 - kw0) >> 1;
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const int iy = iy0 - ty, ix = ix0 - tx;
                if ((unsigned)iy < (unsigned)DL_IH && (unsigned)ix < (unsigned)DL_IW)
                    acc += P_s[(iy * DL_IW + ix) * 16 + (kh0 + 2 * ty) * 4 + kw0 + 2 * tx];
            }
        const long long oidx = n * (DL_OH * DL_OW) + o;      // NCHW, one channel
        const float p = __builtin_amdgcn_rcpf(1.0f + __expf(-acc));
        if (a.logits) a.logits[oidx] = acc;
        if (a.recon) a.recon[oidx] = p;
        if (a.target) {
            const float t = a.target[(long long)n_in_g * (DL_OH * DL_OW) + o];
            const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.0f - p), -100.f);       // BCE log clamp
            loss += -(t * lp + (1.0f - t) * lq);
            const float pq = p * (1.0f - p);
            const float dl = a.coef[g] * (p - t) / fmaxf(pq, 1e-12f) * pq;
            if (a.dlogit) a.dlogit[oidx] = dl;
            dl_s[(oy + 1) * DL_DLW + ox + 1] = dl;
        }
    }
    if (a.loss_sum) {
        loss = wave_sum(loss);
        if (lane == 0) part[wave] = loss;
    }
    __syncthreads();
    if (a.loss_sum && tid == 0) {
        float s = 0.f;
        for (int w = 0; w < DL_WAVES; ++w) s += part[w];
        atomicAdd(a.loss_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + g, s);
    }
    if (!bwd) return;
    // ---- patches[pixel][tap] = dlogit(2iy-1+kh, 2ix-1+kw) as bf16 (P is dead: same buffer); padding rows zero
    for (int px = tid; px < DL_ROWS; px += DL_NTHR) {
        bf16x8 lo, hi;
        if (px < DL_NPIX) {
            const int iy = px / DL_IW, ix = px - iy * DL_IW;
            const float* d0 = dl_s + (2 * iy) * DL_DLW + 2 * ix;         // (2iy-1+kh) + 1 halo row, (2ix-1+kw) + 1 halo column
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                lo[kw] = (bf16)d0[kw]; lo[4 + kw] = (bf16)d0[DL_DLW + kw];
                hi[kw] = (bf16)d0[2 * DL_DLW + kw]; hi[4 + kw] = (bf16)d0[3 * DL_DLW + kw];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { lo[j] = (bf16)0.f; hi[j] = (bf16)0.f; }
        }
        *reinterpret_cast<bf16x8*>(pat_s + px * DL_PP) = lo;
        *reinterpret_cast<bf16x8*>(pat_s + px * DL_PP + 16) = hi;
    }
    __syncthreads();
    // ---- input gradient: one MFMA per 32 pixels, then the conv kernels' accumulator epilogue (lane = channel)
    {
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            a.db, 0, (int)((size_t)a.bwd_groups * a.B * DL_NPIX * DL_C * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16*>(a.r), 0, (int)((size_t)a.bwd_groups * a.B * DL_NPIX * DL_C * 2), 0x00020000);
        const float2 af = aff_s[r], mr = mr_s[r];
        float s1 = 0.f, s2 = 0.f;
        char* const scr = scr_s + wave * 2560;
        for (int t = wave; t < DL_TILES; t += DL_WAVES) {
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            const bf16x8 pa = *reinterpret_cast<const bf16x8*>(pat_s + (t * 32 + r) * DL_PP + h * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, wd, acc, 0, 0, 0);
            // this lane's two output vectors: pixels (lane&15) and 16 + (lane&15) of the tile, channel octet lane>>4
            const unsigned base = (unsigned)((n * DL_NPIX + t * 32) * (DL_C * 2)) + (unsigned)((lane >> 4) * 16);
            const int p0 = t * 32 + (lane & 15), p1 = p0 + 16;
            const unsigned off0 = p0 < DL_NPIX ? base + (unsigned)((lane & 15) * (DL_C * 2)) : 0x40000000u;
            const unsigned off1 = p1 < DL_NPIX ? base + (unsigned)((16 + (lane & 15)) * (DL_C * 2)) : 0x40000000u;
            const int left = DL_NPIX - t * 32;
            if (left >= 32) cr_epilogue_tile<1, false>(acc, scr, lane, 32, off0, off1, orsrc, rrsrc, af.x, af.y, mr.x, mr.y, s1, s2, true);
            else cr_epilogue_tile<1, true>(acc, scr, lane, left, off0, off1, orsrc, rrsrc, af.x, af.y, mr.x, mr.y, s1, s2, true);
        }
        const float t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
        if (h == 0) {
            const int slot = (int)(blockIdx.x + wave) % MMVAE_STAT_SLOTS;
            float2* d = a.red + ((size_t)g * MMVAE_STAT_SLOTS + slot) * DL_C + r;
            atomicAdd(&d->x, t1);
            atomicAdd(&d->y, t2);
        }
    }
    __syncthreads();                                         // the epilogue scratch becomes the waves' dW partials
    // ---- weight gradient dW[ch][tap] = A^T . patches over the 640 pixel rows: 40 k-steps of 16 rows shared by the waves
    {
        const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int ks = wave; ks < DL_ROWS / 16; ks += DL_WAVES) {
            const int row = ks * 16 + 8 * h + q;
            const char* a0 = A_s + row * DL_AP + (16 * (g4 & 1) + 4 * p) * 2;
            const char* b0 = pat_s + row * DL_PP + (4 * p) * 2;          // taps 0..15 for both column halves (16..31 unused)
            const bf16x8 af2 = tr_pair_d(a0, a0 + 4 * DL_AP);
            const bf16x8 bf2 = tr_pair_d(b0, b0 + 4 * DL_PP);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2, bf2, acc, 0, 0, 0);
        }
        float* wred = reinterpret_cast<float*>(scr_s) + wave * (DL_C * 16);
        if (r < 16) {                                        // lane = tap, register j = channel (j&3) + 8*(j>>2) + 4h
#pragma unroll
            for (int j = 0; j < 16; ++j) wred[((j & 3) + 8 * (j >> 2) + 4 * h) * 16 + r] = acc[j];
        }
    }
    __syncthreads();
    {
        const float* wred = reinterpret_cast<const float*>(scr_s);
        float* dst = a.wslab + ((size_t)g * gridDim.x + blockIdx.x) * DL_C * 16;
        for (int i = tid; i < DL_C * 16; i += DL_NTHR) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < DL_WAVES; ++w) s += wred[w * DL_C * 16 + i];
            dst[i] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// CelebA tail (celeba/model.py:146-150): ConvTranspose2d(32, 3, 4, 2, 1) on 32x32 maps -> 64x64x3 + sigmoid + BCE, its input
// and weight gradients.  Same three products as above with j = tap*3 + co (48 of 64 columns) as the narrow dimension; one
// workgroup (8 waves, 124 KB of LDS) owns one image and walks it in 4 strips of SR = 8 input rows so that P (fp32) fits: a
// strip stages input rows y0-1 .. y0+SR (BatchNorm + Swish on the way), makes P for those SR+2 rows, the logits of output rows
// 2y0-1 .. 2y0+2SR (the outer two only for the gradient windows of the strip's own pixels; loss / outputs for the 2SR owned
// rows), the patches and the input gradient of its SR rows; the weight-gradient accumulators stay in registers across the
// strips.  (SR = 4 with 4 waves and two 72 KB workgroups per CU: 334 us in the CelebA step against 245 -- the halo rows are
// half of what a strip stages and the per-strip barriers double.)
constexpr int CL_IH = 32, CL_IW = 32, CL_OH = 64, CL_OW = 64, CL_C = 32, CL_CO = 3, CL_J = 48, CL_SR = 8, CL_NS = CL_IH / CL_SR;
constexpr int CL_NPIX = CL_IH * CL_IW;
constexpr int CL_AROWS = CL_SR + 2, CL_APX = CL_AROWS * CL_IW, CL_ATILES = CL_APX / 32;       // staged rows / pixels / 32-pixel tiles
constexpr int CL_OPX = CL_SR * CL_IW, CL_OTILES = CL_OPX / 32;                                  // owned pixels of a strip
constexpr int CL_AP = 80;              // A: bytes per pixel (32 bf16 + 16)
constexpr int CL_PF = 49;              // P: floats per pixel (48 + 1: the overlap-add reads a column across pixels)
constexpr int CL_PP = 144;             // patches: bytes per pixel (64 bf16 columns, 48 used, + 16)
constexpr int CL_DLR = 2 * CL_SR + 2, CL_DLW = CL_OW + 2;                                        // dlogit rows of a strip, width with halo
constexpr int CL_WAVES = 8, CL_NTHR = CL_WAVES * 64;
constexpr int CL_OFF_A = 0;
constexpr int CL_OFF_P = CL_OFF_A + CL_APX * CL_AP;                       // P fp32 [320][49]; then patches [256][144]; at the end dW partials
constexpr int CL_P_BYTES = CL_APX * CL_PF * 4;
constexpr int CL_OFF_DL = CL_OFF_P + (CL_P_BYTES + 15) / 16 * 16;         // dlogit fp32 [3][18][66]
constexpr int CL_OFF_SCR = CL_OFF_DL + (CL_CO * CL_DLR * CL_DLW * 4 + 15) / 16 * 16;    // per-wave epilogue scratch [8][2560]
constexpr int CL_OFF_TAB = CL_OFF_SCR + CL_WAVES * 2560;
constexpr int CL_LDS = CL_OFF_TAB + 2 * CL_C * 8;
static_assert(CL_OPX * CL_PP <= CL_P_BYTES && CL_WAVES * CL_C * CL_J * 4 <= CL_P_BYTES, "patches / dW partials reuse the P buffer");
static_assert(CL_LDS <= 160 * 1024 - 64, "LDS budget");
static_assert(CL_OTILES == CL_WAVES, "one owned 32-pixel tile per wave");

__global__ __launch_bounds__(CL_NTHR) void dec_last_ca_kernel(const DecLastFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const A_s = smem + CL_OFF_A;
    float* const P_s = reinterpret_cast<float*>(smem + CL_OFF_P);
    char* const pat_s = smem + CL_OFF_P;
    float* const dl_s = reinterpret_cast<float*>(smem + CL_OFF_DL);
    char* const scr_s = smem + CL_OFF_SCR;
    float2* const aff_s = reinterpret_cast<float2*>(smem + CL_OFF_TAB);
    float2* const mr_s = aff_s + CL_C;
    __shared__ float part[CL_WAVES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n_in_g = blockIdx.x, g = blockIdx.y;
    const long long n = (long long)g * a.B + n_in_g;
    const BnFinalizeArgs& f = a.fin;
    const bool bwd = g < a.bwd_groups;

    if (tid < CL_C) {
        float2 aff, mr;
        bn_channel_tables(f, g, tid, aff, mr);
        aff_s[tid] = aff; mr_s[tid] = mr;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {                // the tables backward reads + running statistics: once
        for (int i = tid; i < f.G * CL_C; i += CL_NTHR) {
            float2 aff, mr;
            bn_channel_tables(f, i / CL_C, i % CL_C, aff, mr);
            f.affine[i] = aff; f.meanrstd[i] = mr;
        }
        bn_running_update(f, tid, CL_NTHR);
    }
    // weight fragments, fp32 (32, 3, 4, 4) -> bf16, column j = tap*3 + co:
    //   forward B[k = ch][j] for the two k-steps and the two column tiles; input gradient B[k = j][ch] for the three k-steps
    bf16x8 wf[2][2], wd[3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int j = nt * 32 + r, tap = j / CL_CO, co = j - tap * CL_CO;
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[ks][nt][e] = j < CL_J ? (bf16)a.w[((ks * 16 + 8 * h + e) * CL_CO + co) * 16 + tap] : (bf16)0.f;
        }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = ks * 16 + 8 * h + e, tap = j / CL_CO, co = j - tap * CL_CO;
            wd[ks][e] = (bf16)a.w[(r * CL_CO + co) * 16 + tap];
        }
    __syncthreads();
    const int cv = tid & 3;                                  // CL_NTHR % 4 == 0: a thread keeps its channel octet
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float2 t = aff_s[cv * 8 + e]; sc[e] = t.x; sh[e] = t.y; }
    const float2 af = aff_s[r], mr = mr_s[r];

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.db, 0, (int)((size_t)a.bwd_groups * a.B * CL_NPIX * CL_C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.r), 0, (int)((size_t)a.bwd_groups * a.B * CL_NPIX * CL_C * 2), 0x00020000);
    f32x16 accw[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[nt][e] = 0.f;
    float s1 = 0.f, s2 = 0.f, loss = 0.f;
    const bf16* const rimg = a.r + (size_t)n * CL_NPIX * CL_C;
    constexpr int NV = CL_APX * 4, IT = (NV + CL_NTHR - 1) / CL_NTHR;

    constexpr int NO = CL_CO * CL_DLR * CL_OW;               // logits of a strip
    constexpr int ND = CL_CO * CL_DLR * CL_DLW, DT = (ND + CL_NTHR - 1) / CL_NTHR;     // dlogit buffer elements / per thread
    i32x4c rv[IT];
    auto fetch_rows = [&](int y0) {                          // raw rows y0-1 .. y0+8 of the image (outside: zero)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = tid + it * CL_NTHR, px = v >> 2, iy = y0 - 1 + px / CL_IW;
            rv[it] = i32x4c{0, 0, 0, 0};
            if (v < NV && (unsigned)iy < (unsigned)CL_IH)
                rv[it] = *reinterpret_cast<const i32x4c*>(rimg + ((size_t)(iy * CL_IW + (px & (CL_IW - 1))) * CL_C + cv * 8));
        }
    };
    fetch_rows(0);
    const float* const timg = a.target ? a.target + (size_t)n_in_g * CL_CO * CL_OH * CL_OW : nullptr;
    // targets of a strip's 18 output rows: fetched one strip ahead, parked in the dlogit buffer (halo columns / rows outside the
    // image: zero) where the logits pass picks them up and leaves the gradient -- no global load inside that loop
    float tv[DT];
    auto fetch_targets = [&](int y0) {
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            const int i = tid + k * CL_NTHR;
            const int co = i / (CL_DLR * CL_DLW), rem = i - co * (CL_DLR * CL_DLW);
            const int lo = rem / CL_DLW, c = rem - lo * CL_DLW, oy = 2 * y0 - 1 + lo;
            tv[k] = 0.f;
            if (timg && i < ND && c >= 1 && c <= CL_OW && (unsigned)oy < (unsigned)CL_OH) tv[k] = timg[(co * CL_OH + oy) * CL_OW + c - 1];
        }
    };
    fetch_targets(0);

    for (int strip = 0; strip < CL_NS; ++strip) {
        const int y0 = strip * CL_SR;
        // ---- stage rows y0-1 .. y0+8 (fetched during the previous strip): BatchNorm + Swish -> A (rows outside the image: zero)
        __syncthreads();                                     // the previous strip's readers of A / patches / dlogit are done
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            const int i = tid + k * CL_NTHR;
            if (i < ND) dl_s[i] = tv[k];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = tid + it * CL_NTHR, px = v >> 2, iy = y0 - 1 + px / CL_IW;
            if (v < NV) {
                bf16x8 o;
                if ((unsigned)iy < (unsigned)CL_IH) {
                    const bf16x8 x = __builtin_bit_cast(bf16x8, rv[it]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)swish_fast((float)x[e] * sc[e] + sh[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
                }
                *reinterpret_cast<bf16x8*>(A_s + px * CL_AP + cv * 16) = o;
            }
        }
        if (strip + 1 < CL_NS) { fetch_rows(y0 + CL_SR); fetch_targets(y0 + CL_SR); }   // next strip's: in flight through the rest of this one
        __syncthreads();
        // ---- forward: P[pixel][j] for the 10 staged rows
        if (!(a.dbg & 16))
        for (int t = wave; t < CL_ATILES; t += CL_WAVES) {
            const char* ap = A_s + (t * 32 + r) * CL_AP + h * 16;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap), a1 = *reinterpret_cast<const bf16x8*>(ap + 32);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[0][nt], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[1][nt], acc, 0, 0, 0);
                const int j = nt * 32 + r;
                if (j < CL_J) {                              // lane = column j, register e = pixel (e&3) + 8*(e>>2) + 4h of the tile
#pragma unroll
                    for (int e = 0; e < 16; ++e) P_s[(t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CL_PF + j] = acc[e];
                }
            }
        }
        __syncthreads();
        // ---- logits of output rows 2y0-1 .. 2y0+16 by overlap-add, sigmoid, BCE and its gradient
#pragma unroll 2
        for (int o = (a.dbg & 1) ? NO : tid; o < NO; o += CL_NTHR) {
            const int co = o / (CL_DLR * CL_OW), rem = o - co * (CL_DLR * CL_OW);
            const int lo = rem / CL_OW, ox = rem - lo * CL_OW, oy = 2 * y0 - 1 + lo;
            if ((unsigned)oy >= (unsigned)CL_OH) continue;
            const int kh0 = (oy + 1) & 1, kw0 = (ox + 1) & 1;
            const int iy0 = (oy + 1 - kh0) >> 1, ix0 = (ox + 1 - kw0) >> 1;
            float acc = 0.f;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    // (rows outside the image are staged as zeros, i.e. P = 0 there; a column outside is masked, not branched on)
                    const int iy = iy0 - ty, ix = ix0 - tx, ixc = min(max(ix, 0), CL_IW - 1);
                    const float pv = P_s[((iy - y0 + 1) * CL_IW + ixc) * CL_PF + ((kh0 + 2 * ty) * 4 + kw0 + 2 * tx) * CL_CO + co];
                    acc += ix == ixc ? pv : 0.f;
                }
            const bool owned = lo >= 1 && lo <= 2 * CL_SR;
            const long long oidx = ((n * CL_CO + co) * CL_OH + oy) * CL_OW + ox;       // NCHW
            const float p = __builtin_amdgcn_rcpf(1.0f + __expf(-acc));
            if (owned) {
                if (a.logits) a.logits[oidx] = acc;
                if (a.recon) a.recon[oidx] = p;
            }
            if (a.target) {
                const float t = dl_s[(co * CL_DLR + lo) * CL_DLW + ox + 1];
                const float pq = p * (1.0f - p);
                const float dl = a.coef[g] * (p - t) / fmaxf(pq, 1e-12f) * pq;
                if (owned) {
                    const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.0f - p), -100.f);   // BCE log clamp
                    loss += -(t * lp + (1.0f - t) * lq);
                    if (a.dlogit) a.dlogit[oidx] = dl;
                }
                dl_s[(co * CL_DLR + lo) * CL_DLW + ox + 1] = dl;
            }
        }
        if (!bwd) continue;                                  // (uniform over the workgroup)
        __syncthreads();
        // ---- patches[pixel][j] = dlogit(co, 2iy-1+kh, 2ix-1+kw) as bf16 for the 8 owned rows (P is dead: same buffer);
        //      two threads per pixel (kernel rows 0-1 / 2-3), the second also zeroes the 16 unused columns
        if (!(a.dbg & 2)) {
            const int px = tid >> 1, half = tid & 1;
            const int iyl = px / CL_IW, ix = px - iyl * CL_IW;
            bf16 v[24];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw)
#pragma unroll
                    for (int co = 0; co < CL_CO; ++co)
                        v[(kk * 4 + kw) * CL_CO + co] = (bf16)dl_s[(co * CL_DLR + 2 * iyl + 2 * half + kk) * CL_DLW + 2 * ix + kw];
            char* dst = pat_s + px * CL_PP + half * 48;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[q * 8 + e];
                *reinterpret_cast<bf16x8*>(dst + q * 16) = o;
            }
            if (half) {
                const i32x4c z = {0, 0, 0, 0};
                *reinterpret_cast<i32x4c*>(pat_s + px * CL_PP + 96) = z;
                *reinterpret_cast<i32x4c*>(pat_s + px * CL_PP + 112) = z;
            }
        }
        __syncthreads();
        // ---- input gradient of the wave's 32 owned pixels: three MFMAs, then the conv kernels' accumulator epilogue
        if (!(a.dbg & 4)) {
            const int t = wave;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const bf16x8 pa = *reinterpret_cast<const bf16x8*>(pat_s + (t * 32 + r) * CL_PP + ks * 32 + h * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, wd[ks], acc, 0, 0, 0);
            }
            const unsigned base = (unsigned)((n * CL_NPIX + y0 * CL_IW + t * 32) * (CL_C * 2)) + (unsigned)((lane >> 4) * 16);
            const unsigned off0 = base + (unsigned)((lane & 15) * (CL_C * 2)), off1 = base + (unsigned)((16 + (lane & 15)) * (CL_C * 2));
            cr_epilogue_tile<1, false>(acc, scr_s + wave * 2560, lane, 32, off0, off1, orsrc, rrsrc, af.x, af.y, mr.x, mr.y, s1, s2, true);
        }
        // ---- weight gradient dW[ch][j] += A^T . patches over the 256 owned pixels: 16 k-steps of 16 rows, two per wave
        if (!(a.dbg & 8)) {
            const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
            for (int ks = wave; ks < CL_OPX / 16; ks += CL_WAVES) {
                const int row = ks * 16 + 8 * h + q;
                const char* a0 = A_s + (row + CL_IW) * CL_AP + (16 * (g4 & 1) + 4 * p) * 2;       // owned pixels start at staged row 1
                const bf16x8 af2 = tr_pair_d(a0, a0 + 4 * CL_AP);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const char* b0 = pat_s + row * CL_PP + (nt * 32 + 16 * (g4 & 1) + 4 * p) * 2;
                    const bf16x8 bf2 = tr_pair_d(b0, b0 + 4 * CL_PP);
                    accw[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af2, bf2, accw[nt], 0, 0, 0);
                }
            }
        }
    }
    // ---- loss, BatchNorm-backward sums, weight-gradient partial of the image
    if (a.loss_sum) {
        loss = wave_sum(loss);
        if (lane == 0) part[wave] = loss;
    }
    __syncthreads();
    if (a.loss_sum && tid == 0) {
        float s = 0.f;
        for (int w = 0; w < CL_WAVES; ++w) s += part[w];
        atomicAdd(a.loss_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + g, s);
    }
    if (!bwd) return;
    {
        const float t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
        if (h == 0) {
            const int slot = (int)(blockIdx.x + wave) % MMVAE_STAT_SLOTS;
            float2* d = a.red + ((size_t)g * MMVAE_STAT_SLOTS + slot) * CL_C + r;
            atomicAdd(&d->x, t1);
            atomicAdd(&d->y, t2);
        }
    }
    float* const wred = P_s;                                 // [8 waves][32 ch][48]  (every reader of the patches passed the barrier above)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = nt * 32 + r;
        if (j < CL_J) {                                      // lane = column j, register e = channel (e&3) + 8*(e>>2) + 4h
#pragma unroll
            for (int e = 0; e < 16; ++e) wred[(wave * CL_C + (e & 3) + 8 * (e >> 2) + 4 * h) * CL_J + j] = accw[nt][e];
        }
    }
    __syncthreads();
    float* dst = a.wslab + ((size_t)g * gridDim.x + blockIdx.x) * CL_C * CL_J;
    for (int i = tid; i < CL_C * CL_J; i += CL_NTHR) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < CL_WAVES; ++w) s += wred[w * CL_C * CL_J + i];
        dst[i] = s;
    }
}

}  // namespace

bool dec_last_ca_applies(const DecLastFusedArgs& a) {
    return a.Cin == CL_C && a.Cout == CL_CO && a.IH == CL_IH && a.IW == CL_IW && a.act == ACT_SWISH && mmvae_knob("dec_last_ca", 1) != 0 &&
           (size_t)a.G * a.B * CL_NPIX * CL_C * 2 < 0x40000000ull;
}
int launch_dec_last_ca(const DecLastFusedArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(dec_last_ca_applies(a), "dec_last_ca: not the 32x32x32 -> 64x64x3 tail");
    MMVAE_REQUIRE(a.r && a.w && a.G >= 1 && a.B >= 1 && a.bwd_groups >= 0 && a.bwd_groups <= a.G, "dec_last_ca: arguments");
    MMVAE_REQUIRE(a.bwd_groups == 0 || (a.target && a.db && a.red && a.wslab), "dec_last_ca: backward outputs missing");
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_last_ca_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CL_LDS);
    DecLastFusedArgs b = a;
    b.dbg = mmvae_knob("dec_last_ca_dbg", 0);      // measurement aid: bit 0 no logits, 1 no patches, 2 no input gradient, 3 no weight gradient, 4 no P
    MMVAE_LAUNCH(dec_last_ca_kernel, dim3(a.B, a.G), dim3(CL_NTHR), CL_LDS, s, b);
    return mmvae_check_launch("dec_last_ca");
}

// strips = 1: one workgroup per image
bool dec_last_mfma_applies(const DecLastFusedArgs& a) {
    return a.Cin == 32 && a.Cout == 1 && a.IH == DL_IH && a.IW == DL_IW && a.act == ACT_SWISH && mmvae_knob("dec_last_mfma", 1) != 0 &&
           (size_t)a.G * a.B * DL_NPIX * DL_C * 2 < 0x40000000ull;
}
int launch_dec_last_mfma(const DecLastFusedArgs& a, hipStream_t s) {
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_last_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DL_LDS);   // (+ 32 B static)
    MMVAE_LAUNCH(dec_last_mfma_kernel, dim3(a.B, a.G), dim3(DL_NTHR), DL_LDS, s, a);
    return mmvae_check_launch("dec_last_mfma");
}
