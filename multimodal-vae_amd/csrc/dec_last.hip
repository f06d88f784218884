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

}  // namespace

// strips = 1: one workgroup per image
bool dec_last_mfma_applies(const DecLastFusedArgs& a) {
    return a.Cin == 32 && a.IH == DL_IH && a.IW == DL_IW && a.act == ACT_SWISH && mmvae_knob("dec_last_mfma", 1) != 0 &&
           (size_t)a.G * a.B * DL_NPIX * DL_C * 2 < 0x40000000ull;
}
int launch_dec_last_mfma(const DecLastFusedArgs& a, hipStream_t s) {
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_last_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DL_LDS);   // (+ 32 B static)
    MMVAE_LAUNCH(dec_last_mfma_kernel, dim3(a.B, a.G), dim3(DL_NTHR), DL_LDS, s, a);
    return mmvae_check_launch("dec_last_mfma");
}
