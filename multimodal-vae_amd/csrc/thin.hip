// Direct (non-MFMA) kernels for the thin last layer of the image decoder: ConvTranspose2d(Cin, COUT<=3, k4, s2, p1)
// (multimnist/model.py:208, celeba/model.py:150).  With one or three output channels an MFMA tile would be >90 %
// padding, so these are bandwidth-shaped VALU kernels: 16-B activation loads, fp32 weights broadcast from LDS,
// and the sigmoid / BCE / d-activation / BatchNorm-backward reductions fused in.
#include "thin.h"
#include "bn_dev.h"
#include <math.h>

namespace {

constexpr int TPB = 256;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float acc) {
    // 8 bf16 MACs as 4 packed v_dot2c_f32_bf16 (fp32 accumulate)
    union { bf16x8 v; bf16x2 p[4]; } ua, ub;
    ua.v = a; ub.v = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(ua.p[i], ub.p[i], acc, false);
    return acc;
}

// forward: one workgroup per (image, strip of FR input rows); the activated input strip (+1 halo row each side,
// zero-filled outside the image) is staged once in LDS, so every input pixel is read from global memory once.
// Weights sit in LDS as bf16 [kh][kw][co][32 ci]; each output pixel is 4 taps x 32 channels = 16 dot2 x 4 per
// output channel.  Fused sigmoid + BCE (+ gradient).
constexpr int FR = 5;                       // input rows per strip (25 = 5*5, 32 = 6*5+2: the tail strip is masked)
template <int COUT, int CIN>
__global__ __launch_bounds__(TPB) void convt_last_fwd_kernel(const ConvTLastFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* wl = reinterpret_cast<bf16*>(smem);                       // [16 taps][COUT][CIN]
    bf16* tile = wl + 16 * COUT * CIN;                              // [(FR+2)][IW+2][CIN]
    __shared__ float part[TPB / 64];
    const int strips = (a.IH + FR - 1) / FR;
    const int n_in_g = blockIdx.x / strips, strip = blockIdx.x - n_in_g * strips;
    const int g = blockIdx.y;
    const long long n = (long long)g * a.B + n_in_g;
    const int iy_base = strip * FR - 1;                             // first staged row (halo)
    const int TW = a.IW + 2;
    for (int i = threadIdx.x; i < 16 * COUT * CIN; i += TPB) {
        const int ci = i % CIN, co = (i / CIN) % COUT, tap = i / (CIN * COUT);
        wl[i] = (bf16)a.w[(ci * COUT + co) * 16 + tap];            // weight (Cin, Cout, 4, 4)
    }
    constexpr int VPP = CIN / 8;                                    // 16-byte vectors per pixel
    const int nvec = (FR + 2) * TW * VPP;
    for (int v = threadIdx.x; v < nvec; v += TPB) {
        const int q = v % VPP, pixl = v / VPP;
        const int ty = pixl / TW, tx = pixl - ty * TW;
        const int iy = iy_base + ty, ix = tx - 1;
        const bool ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        const long long src = ok ? ((n * a.IH + iy) * a.IW + ix) * a.Cin + q * 8 : 0;
        bf16x8 val = *reinterpret_cast<const bf16x8*>(a.act + src);
        if (!ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(tile + (size_t)pixl * CIN + q * 8) = val;
    }
    __syncthreads();
    const int OH = 2 * a.IH, OW = 2 * a.IW;
    const int oy_lo = 2 * strip * FR;
    const int nout = 2 * FR * OW;
    float loss = 0.f;
    for (int o = threadIdx.x; o < nout; o += TPB) {
        const int oyl = o / OW, ox = o - oyl * OW;
        const int oy = oy_lo + oyl;
        if (oy >= OH) continue;
        const int kh0 = (oy + 1) & 1, kw0 = (ox + 1) & 1;
        const int iy0 = (oy + 1 - kh0) >> 1, ix0 = (ox + 1 - kw0) >> 1;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const int iy = iy0 - ty, ix = ix0 - tx;             // in [-1, IH] x [-1, IW]: inside the halo tile
                const bf16* src = tile + ((size_t)(iy - iy_base) * TW + (ix + 1)) * CIN;
                const bf16* wp = wl + ((kh0 + 2 * ty) * 4 + kw0 + 2 * tx) * COUT * CIN;
#pragma unroll
                for (int c0 = 0; c0 < CIN; c0 += 8) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c0);
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
                        acc[co] = dot8(v, *reinterpret_cast<const bf16x8*>(wp + co * CIN + c0), acc[co]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const long long oidx = ((n * COUT + co) * OH + oy) * OW + ox;         // NCHW
            const float l = acc[co];
            // F.sigmoid and the two BCE logs on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1e-7
            // relative): the accurate library versions were half of this kernel's VALU instructions
            const float p = __builtin_amdgcn_rcpf(1.0f + __expf(-l));
            if (a.logits) a.logits[oidx] = l;
            if (a.recon) a.recon[oidx] = p;
            if (a.target) {
                const float t = a.target[(((long long)n_in_g * COUT + co) * OH + oy) * OW + ox];
                const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.0f - p), -100.f);   // BCE log clamp
                loss += -(t * lp + (1.0f - t) * lq);
                if (a.dlogit) {
                    const float pq = p * (1.0f - p);
                    a.dlogit[oidx] = a.coef[g] * (p - t) / fmaxf(pq, 1e-12f) * pq;
                }
            }
        }
    }
    if (a.loss_sum) {
        loss = wave_sum(loss);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = loss;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < TPB / 64; ++w) s += part[w];
            atomicAdd(a.loss_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + g, s);
        }
    }
}

// backward-data: four threads per INPUT pixel (8 of the 32 channels each, one 16-byte vector).  The 16*COUT upstream
// values of the pixel are packed to bf16 pairs once; per channel the reduction over (co, tap) is 8*COUT dot2 against
// bf16 weights [ci][co][16 taps] read from LDS as 16-byte vectors.  Fused d-activation of the producer layer + its two
// BatchNorm-backward sums (lanes with equal lane&3 own the same channels: xor-shuffles over the other lane bits,
// one LDS pass across the waves, one atomic per channel per workgroup into a slot).
template <int COUT>
__global__ __launch_bounds__(TPB) void convt_last_dgrad_kernel(const ConvTLastDgradArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 wl[32 * COUT * 16];          // [ci][co][tap]
    __shared__ float2 red_s[TPB / 64][32];
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 32 * COUT * 16; i += TPB) {
        const int tap = i & 15, co = (i >> 4) % COUT, ci = i / (16 * COUT);
        wl[i] = (bf16)a.w[(ci * COUT + co) * 16 + tap];
    }
    __syncthreads();
    const int OH = 2 * a.IH, OW = 2 * a.IW;
    const long long per_group = (long long)a.B * a.IH * a.IW;
    const int cg = threadIdx.x & 3;                   // channel group: channels 8*cg .. 8*cg+7
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (long long i = (long long)blockIdx.x * (TPB / 4) + (threadIdx.x >> 2); i < per_group; i += (long long)gridDim.x * (TPB / 4)) {
        const int ix = (int)(i % a.IW);
        long long t1 = i / a.IW;
        const int iy = (int)(t1 % a.IH);
        const long long n = (long long)g * a.B + t1 / a.IH;
        // upstream values: unconditional loads from clamped addresses, out-of-image taps zeroed, packed to bf16 pairs
        bf16x8 dlp[2 * COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int oy = 2 * iy - 1 + kh;
                const int oyc = min(max(oy, 0), OH - 1);
                const float* row = a.dlogit + ((n * COUT + co) * OH + oyc) * OW;
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ox = 2 * ix - 1 + kw;
                    const bool ok = (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
                    const float d = row[min(max(ox, 0), OW - 1)];
                    dlp[co * 2 + (kh >> 1)][(kh & 1) * 4 + kw] = (bf16)(ok ? d : 0.f);
                }
            }
        const long long pix = (n * a.IH + iy) * a.IW + ix;
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.r + pix * a.Cin + cg * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            float acc = 0.f;
            const bf16* wp = wl + c * COUT * 16;
#pragma unroll
            for (int q = 0; q < 2 * COUT; ++q) acc = dot8(dlp[q], *reinterpret_cast<const bf16x8*>(wp + q * 8), acc);
            const float rr = (float)rv[j];
            const float2 af = a.affine[g * a.Cin + c];
            const float2 mr = a.meanrstd[g * a.Cin + c];
            const float v = acc * act_bwd(a.act, rr * af.x + af.y);
            s1[j] += v;
            s2[j] += v * (rr - mr.x) * mr.y;
            o[j] = (bf16)v;
        }
        *reinterpret_cast<bf16x8*>(a.db + pix * a.Cin + cg * 8) = o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = s1[j], y = s2[j];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
        if ((threadIdx.x & 63) < 4) red_s[threadIdx.x >> 6][cg * 8 + j] = make_float2(x, y);
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float x = 0.f, y = 0.f;
        for (int w = 0; w < TPB / 64; ++w) { x += red_s[w][threadIdx.x].x; y += red_s[w][threadIdx.x].y; }
        const int slot = blockIdx.x % MMVAE_STAT_SLOTS;
        atomicAdd(&a.red[(g * MMVAE_STAT_SLOTS + slot) * a.Cin + threadIdx.x].x, x);
        atomicAdd(&a.red[(g * MMVAE_STAT_SLOTS + slot) * a.Cin + threadIdx.x].y, y);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused tail of the image decoder (see thin.h: DecLastFusedArgs).  One workgroup per (image, strip of FR input rows).
template <int CIN>
__global__ __launch_bounds__(TPB, 4) void dec_last_fused_kernel(const DecLastFusedArgs a) {
    static_assert(CIN == 32, "the fused tail is written for 32 input channels");
    constexpr int VPP = CIN / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TW = a.IW + 2, OW = 2 * a.IW, OH = 2 * a.IH, DLW = OW + 2;
    bf16* wl = reinterpret_cast<bf16*>(smem);                       // [16 taps][CIN] bf16 (forward dot products)
    float* wf = reinterpret_cast<float*>(wl + 16 * CIN);            // [16 taps][CIN] fp32 (input gradient)
    float2* aff_s = reinterpret_cast<float2*>(wf + CIN * 16);       // [CIN]
    float2* mr_s = aff_s + CIN;                                     // [CIN]
    float* dl_s = reinterpret_cast<float*>(mr_s + CIN);             // [2*FR + 2][DLW] dlogit with a zero halo
    float* tg_s = dl_s + (2 * FR + 2) * DLW;                        // [2*FR + 2][OW] BCE targets of the strip's rows (+ halo)
    bf16* tile = reinterpret_cast<bf16*>(tg_s + (2 * FR + 2) * OW);  // [(FR+2)][TW][CIN] activated input strip
    bf16* raw_s = tile + (FR + 2) * TW * CIN;                       // same shape: the raw (pre-BatchNorm) values
    float* wred = reinterpret_cast<float*>(raw_s);                  // [4 waves][CIN][16] weight-gradient partials (after the
                                                                    // last read of raw_s: 8 KB <= (FR+2)*TW*CIN*2 B)
    __shared__ float part[TPB / 64];
    __shared__ float2 red_s[TPB / 64][CIN];
    const int strips = (a.IH + FR - 1) / FR;
    const int n_in_g = blockIdx.x / strips, strip = blockIdx.x - n_in_g * strips;
    const int g = blockIdx.y;
    const long long n = (long long)g * a.B + n_in_g;
    const int iy_base = strip * FR - 1;
    const int tid = threadIdx.x;
    const BnFinalizeArgs& f = a.fin;

    // ---- BatchNorm tables of this group; block (0,0): tables of every group + running statistics
    if (tid < CIN) {
        float2 aff, mr;
        bn_channel_tables(f, g, tid, aff, mr);
        aff_s[tid] = aff; mr_s[tid] = mr;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = tid; i < f.G * CIN; i += TPB) {
            float2 aff, mr;
            bn_channel_tables(f, i / CIN, i % CIN, aff, mr);
            f.affine[i] = aff; f.meanrstd[i] = mr;
        }
        bn_running_update(f, tid, TPB);
    }
    for (int i = tid; i < 16 * CIN; i += TPB) {
        const int ci = i % CIN, tap = i / CIN;
        const float wv = a.w[ci * 16 + tap];                          // weight (Cin, 1, 4, 4)
        wl[i] = (bf16)wv;
        wf[i] = wv;
    }
    for (int i = tid; i < (2 * FR + 2) * DLW; i += TPB) dl_s[i] = 0.f;
    // every global read of the workgroup is issued in this first phase (strip, weights, statistics, targets): the rest of
    // the kernel runs out of LDS, so a workgroup pays ONE global-memory latency, not one per phase
    const int oy_lo = 2 * strip * FR;
    if (a.target)
        for (int i = tid; i < (2 * FR + 2) * OW; i += TPB) {
            const int oy = oy_lo - 1 + i / OW;
            tg_s[i] = (unsigned)oy < (unsigned)OH ? a.target[((long long)n_in_g * OH + oy) * OW + (i % OW)] : 0.f;
        }
    constexpr int NV_MAX = 4;                                       // (FR+2)*(IW+2)*VPP <= 1024 (checked by the launcher)
    const int nvec = (FR + 2) * TW * VPP;
    bf16x8 rvv[NV_MAX];
#pragma unroll
    for (int i = 0; i < NV_MAX; ++i) {
        const int v = tid + i * TPB;
        const int pixl = v / VPP, q = v % VPP;
        const int ty = pixl / TW, tx = pixl - ty * TW;
        const int iy = iy_base + ty, ix = tx - 1;
        const bool ok = v < nvec && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        const long long src = ok ? ((n * a.IH + iy) * a.IW + ix) * CIN + q * 8 : 0;
        rvv[i] = *reinterpret_cast<const bf16x8*>(a.r + src);
        if (!ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) rvv[i][j] = (bf16)0.f;
        }
    }
    __syncthreads();
    // ---- stage the strip: raw -> BatchNorm affine -> Swish -> bf16 (zero outside the image: the conv pads activations)
#pragma unroll
    for (int i = 0; i < NV_MAX; ++i) {
        const int v = tid + i * TPB;
        if (v >= nvec) continue;
        const int pixl = v / VPP, q = v % VPP;
        const int ty = pixl / TW, tx = pixl - ty * TW;
        const int iy = iy_base + ty, ix = tx - 1;
        const bool ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        bf16x8 val;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 af = aff_s[q * 8 + j];
            val[j] = ok ? (bf16)act_fwd(a.act, (float)rvv[i][j] * af.x + af.y) : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(tile + (size_t)pixl * CIN + q * 8) = val;
        *reinterpret_cast<bf16x8*>(raw_s + (size_t)pixl * CIN + q * 8) = rvv[i];
    }
    __syncthreads();
    // ---- forward over the strip's output rows plus one halo row above and below (the input gradient needs their dlogit)
    const int nout = (2 * FR + 2) * OW;
    const bool bwd = g < a.bwd_groups;
    float loss = 0.f;
    for (int o = tid; o < nout; o += TPB) {
        const int oyl = o / OW - 1, ox = o - (oyl + 1) * OW;
        const int oy = oy_lo + oyl;
        if ((unsigned)oy >= (unsigned)OH) continue;
        const bool own = oyl >= 0 && oyl < 2 * FR;
        if (!own && !bwd) continue;
        const int kh0 = (oy + 1) & 1, kw0 = (ox + 1) & 1;
        const int iy0 = (oy + 1 - kh0) >> 1, ix0 = (ox + 1 - kw0) >> 1;
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const int iy = iy0 - ty, ix = ix0 - tx;             // inside the halo tile
                const bf16* src = tile + ((size_t)(iy - iy_base) * TW + (ix + 1)) * CIN;
                const bf16* wp = wl + ((kh0 + 2 * ty) * 4 + kw0 + 2 * tx) * CIN;
#pragma unroll
                for (int c0 = 0; c0 < CIN; c0 += 8)
                    acc = dot8(*reinterpret_cast<const bf16x8*>(src + c0), *reinterpret_cast<const bf16x8*>(wp + c0), acc);
            }
        const long long oidx = (n * OH + oy) * OW + ox;              // NCHW, one channel
        const float p = __builtin_amdgcn_rcpf(1.0f + __expf(-acc));
        if (own) {
            if (a.logits) a.logits[oidx] = acc;
            if (a.recon) a.recon[oidx] = p;
        }
        if (a.target) {
            const float t = tg_s[(oyl + 1) * OW + ox];
            if (own) {
                const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.0f - p), -100.f);   // BCE log clamp
                loss += -(t * lp + (1.0f - t) * lq);
            }
            const float pq = p * (1.0f - p);
            const float dl = a.coef[g] * (p - t) / fmaxf(pq, 1e-12f) * pq;
            if (own && a.dlogit) a.dlogit[oidx] = dl;
            if (bwd) dl_s[(oyl + 1) * DLW + ox + 1] = dl;
        }
    }
    if (a.loss_sum) {
        loss = wave_sum(loss);
        if ((tid & 63) == 0) part[tid >> 6] = loss;
    }
    __syncthreads();
    if (a.loss_sum && tid == 0) {
        float s = 0.f;
        for (int w = 0; w < TPB / 64; ++w) s += part[w];
        atomicAdd(a.loss_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + g, s);
    }
    if (!bwd) return;
    // ---- input gradient: (pixel, 8 channels) per thread-item; d-Swish of the producer + BatchNorm-backward sums
    const int rows_here = min(FR, a.IH - strip * FR);
    const int items = rows_here * a.IW * VPP;
    const int cg = tid & 3;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int it = tid; it < items; it += TPB) {
        const int pixl = it >> 2;
        const int iyl = pixl / a.IW, ix = pixl - iyl * a.IW;
        float dl[16];
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) dl[kh * 4 + kw] = dl_s[(2 * iyl + kh) * DLW + 2 * ix + kw];
        const long long pix = (n * a.IH + strip * FR + iyl) * a.IW + ix;
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(raw_s + ((size_t)(iyl + 1) * TW + ix + 1) * CIN + cg * 8);
        bf16x8 o;
        float din[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) din[j] = 0.f;
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {                    // 8 channels of one tap: two 16-byte LDS reads
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wf + t * CIN + cg * 8);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wf + t * CIN + cg * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { din[j] += dl[t] * w0[j]; din[4 + j] += dl[t] * w1[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            const float acc = din[j];
            const float rr = (float)rv[j];
            const float2 af = aff_s[c], mr = mr_s[c];
            const float v = acc * act_bwd(a.act, rr * af.x + af.y);
            s1[j] += v;
            s2[j] += v * (rr - mr.x) * mr.y;
            o[j] = (bf16)v;
        }
        *reinterpret_cast<bf16x8*>(a.db + pix * CIN + cg * 8) = o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = s1[j], y = s2[j];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
        if ((tid & 63) < 4) red_s[tid >> 6][cg * 8 + j] = make_float2(x, y);
    }
    __syncthreads();                                                // raw_s is dead from here on: wred takes its place
    // ---- weight gradient dW[c][tap] = sum over the strip's pixels of act[pix][c] * dlogit[pix @ tap]: one MFMA pair per
    //      wave (wave w owns pixels 32w .. 32w+31 as the k dimension), summed across the waves through LDS
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int fr = lane & 15, fq = lane >> 4;
        const int npix = rows_here * a.IW;
        f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int p0 = wave * 32; p0 < npix; p0 += 128) {
            bf16x8 av[2], bv;
            const int kh = fr >> 2, kw = fr & 3;
            int iyl = (p0 + fq * 8) / a.IW, ix = p0 + fq * 8 - iyl * a.IW;        // walked incrementally over the 8 pixels
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pp = p0 + fq * 8 + j;
                const bool ok = pp < npix;
                if (j > 0 && ++ix == a.IW) { ix = 0; ++iyl; }
                const bf16* ap = tile + ((size_t)((ok ? iyl : 0) + 1) * TW + (ok ? ix : 0) + 1) * CIN;
                av[0][j] = ok ? ap[fr] : (bf16)0.f;
                av[1][j] = ok ? ap[16 + fr] : (bf16)0.f;
                bv[j] = ok ? (bf16)dl_s[(2 * (ok ? iyl : 0) + kh) * DLW + 2 * (ok ? ix : 0) + kw] : (bf16)0.f;
            }
            acc2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv, acc2[0], 0, 0, 0);
            acc2[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv, acc2[1], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) wred[(wave * CIN + ct * 16 + fq * 4 + j) * 16 + fr] = acc2[ct][j];
    }
    __syncthreads();
    if (tid < CIN) {
        float x = 0.f, y = 0.f;
        for (int w = 0; w < TPB / 64; ++w) { x += red_s[w][tid].x; y += red_s[w][tid].y; }
        const int slot = blockIdx.x % MMVAE_STAT_SLOTS;
        atomicAdd(&a.red[(g * MMVAE_STAT_SLOTS + slot) * CIN + tid].x, x);
        atomicAdd(&a.red[(g * MMVAE_STAT_SLOTS + slot) * CIN + tid].y, y);
    }
    {
        float* dst = a.wslab + ((size_t)g * gridDim.x + blockIdx.x) * CIN * 16;
        for (int i = tid; i < CIN * 16; i += TPB)
            dst[i] = wred[i] + wred[CIN * 16 + i] + wred[2 * CIN * 16 + i] + wred[3 * CIN * 16 + i];
    }
}

}  // namespace

int launch_convt_last_fwd(const ConvTLastFwdArgs& a, hipStream_t s) {
    MMVAE_REQUIRE((a.Cin == 32 || (a.Cin == 64 && a.Cout == 3)) && (a.Cout == 1 || a.Cout == 3) && a.G >= 1 && a.G <= 4,
                  "convT last fwd: Cin=%d Cout=%d G=%d", a.Cin, a.Cout, a.G);
    const int strips = (a.IH + FR - 1) / FR;
    dim3 grid(a.B * strips, a.G);
    size_t lds = (size_t)16 * a.Cin * a.Cout * sizeof(bf16) + (size_t)(FR + 2) * (a.IW + 2) * a.Cin * sizeof(bf16);
    if (a.Cin == 64) hipLaunchKernelGGL((convt_last_fwd_kernel<3, 64>), grid, dim3(TPB), lds, s, a);
    else if (a.Cout == 1) hipLaunchKernelGGL((convt_last_fwd_kernel<1, 32>), grid, dim3(TPB), lds, s, a);
    else hipLaunchKernelGGL((convt_last_fwd_kernel<3, 32>), grid, dim3(TPB), lds, s, a);
    return mmvae_check_launch("convt_last_fwd");
}
int launch_convt_last_dgrad(const ConvTLastDgradArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.Cin == 32 && (a.Cout == 1 || a.Cout == 3) && a.G >= 1 && a.G <= 4, "convT last dgrad: Cin=%d Cout=%d", a.Cin, a.Cout);
    long long per_group = (long long)a.B * a.IH * a.IW;
    dim3 grid((unsigned)min((per_group * 4 + TPB - 1) / TPB, (long long)8192), a.G);
    if (a.Cout == 1) hipLaunchKernelGGL(convt_last_dgrad_kernel<1>, grid, dim3(TPB), 0, s, a);
    else hipLaunchKernelGGL(convt_last_dgrad_kernel<3>, grid, dim3(TPB), 0, s, a);
    return mmvae_check_launch("convt_last_dgrad");
}

int dec_last_fused_strips(int IH) { return (IH + FR - 1) / FR; }
int launch_dec_last_fused(const DecLastFusedArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.Cin == 32 && a.G >= 1 && a.G <= 4 && a.bwd_groups >= 0 && a.bwd_groups <= a.G && a.fin.C == a.Cin && a.fin.G >= a.G,
                  "dec last fused: Cin=%d G=%d bwd_groups=%d", a.Cin, a.G, a.bwd_groups);
    MMVAE_REQUIRE(a.bwd_groups == 0 || (a.target && a.db && a.red && a.wslab), "dec last fused: backward outputs missing");
    MMVAE_REQUIRE(!a.fin.training || a.fin.count > 1.f, "Expected more than 1 value per channel when training");
    if (dec_last_mfma_applies(a)) return launch_dec_last_mfma(a, s);
    const int strips = dec_last_fused_strips(a.IH);
    const int TW = a.IW + 2, DLW = 2 * a.IW + 2;
    const size_t lds = (size_t)16 * 32 * sizeof(bf16) + (size_t)32 * 16 * sizeof(float) + 2 * 32 * sizeof(float2) +
                       (size_t)(2 * FR + 2) * DLW * sizeof(float) +
                       (size_t)(2 * FR + 2) * 2 * a.IW * sizeof(float) + (size_t)2 * (FR + 2) * TW * 32 * sizeof(bf16);
    MMVAE_REQUIRE((FR + 2) * TW * 4 <= 1024 && (size_t)(FR + 2) * TW * 32 * sizeof(bf16) >= (size_t)4 * 32 * 16 * sizeof(float),
                  "dec last fused: IW=%d out of range", a.IW);
    MMVAE_LAUNCH(dec_last_fused_kernel<32>, dim3(a.B * strips, a.G), dim3(TPB), lds, s, a);
    return mmvae_check_launch("dec_last_fused");
}
