// Direct (non-MFMA) kernels for the thin last layer of the image decoder: ConvTranspose2d(Cin, COUT<=3, k4, s2, p1)
// (multimnist/model.py:208, celeba/model.py:150).  With one or three output channels an MFMA tile would be >90 %
// padding, so these are bandwidth-shaped VALU kernels: 16-B activation loads, fp32 weights broadcast from LDS,
// and the sigmoid / BCE / d-activation / BatchNorm-backward reductions fused in.
#include "thin.h"
#include <math.h>

namespace {

constexpr int TPB = 256;

// forward: one thread per output pixel (all COUT channels); fused sigmoid + BCE(+gradient)
template <int COUT>
__global__ __launch_bounds__(TPB) void convt_last_fwd_kernel(const ConvTLastFwdArgs a) {
    __shared__ float wl[16 * 32 * COUT];          // [kh][kw][ci][co]
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 16 * a.Cin * COUT; i += TPB) {
        int co = i % COUT, ci = (i / COUT) % a.Cin, tap = i / (COUT * a.Cin);
        wl[i] = a.w[(ci * COUT + co) * 16 + tap];                  // weight (Cin, Cout, 4, 4)
    }
    __syncthreads();
    const int OH = 2 * a.IH, OW = 2 * a.IW;
    const long long per_group = (long long)a.B * OH * OW;
    float loss = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < per_group; i += (long long)gridDim.x * TPB) {
        const int ox = (int)(i % OW);
        long long t1 = i / OW;
        const int oy = (int)(t1 % OH);
        const int nb = (int)(t1 / OH);
        const long long n = (long long)g * a.B + nb;
        const int kh0 = (oy + 1) & 1, kw0 = (ox + 1) & 1;
        const int iy0 = (oy + 1 - kh0) >> 1, ix0 = (ox + 1 - kw0) >> 1;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
            const int iy = iy0 - ty, kh = kh0 + 2 * ty;
            if ((unsigned)iy >= (unsigned)a.IH) continue;
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const int ix = ix0 - tx, kw = kw0 + 2 * tx;
                if ((unsigned)ix >= (unsigned)a.IW) continue;
                const bf16* src = a.act + ((n * a.IH + iy) * a.IW + ix) * a.Cin;
                const float* wp = wl + (kh * 4 + kw) * a.Cin * COUT;
                for (int c0 = 0; c0 < a.Cin; c0 += 8) {
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = (float)v[j];
#pragma unroll
                        for (int co = 0; co < COUT; ++co) acc[co] += x * wp[(c0 + j) * COUT + co];
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const long long o = ((n * COUT + co) * OH + oy) * OW + ox;         // NCHW
            const float l = acc[co];
            const float p = 1.0f / (1.0f + expf(-l));                           // F.sigmoid
            if (a.logits) a.logits[o] = l;
            if (a.recon) a.recon[o] = p;
            if (a.target) {
                const float t = a.target[(((long long)nb * COUT + co) * OH + oy) * OW + ox];
                const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.0f - p), -100.f);   // BCE log clamp
                loss += -(t * lp + (1.0f - t) * lq);
                if (a.dlogit) {
                    const float pq = p * (1.0f - p);
                    a.dlogit[o] = a.coef[g] * (p - t) / fmaxf(pq, 1e-12f) * pq;
                }
            }
        }
    }
    if (a.loss_sum) {
        loss = wave_sum(loss);
        __shared__ float part[TPB / 64];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = loss;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < TPB / 64; ++w) s += part[w];
            atomicAdd(a.loss_sum + g, s);
        }
    }
}

// backward-data: one thread per INPUT pixel, all Cin (<= 32) channels; fused d-activation + BatchNorm-backward sums
template <int COUT>
__global__ __launch_bounds__(TPB) void convt_last_dgrad_kernel(const ConvTLastDgradArgs a) {
    __shared__ float wl[16 * COUT * 32];          // [kh][kw][co][ci]
    __shared__ float2 red_s[TPB / 64][32];
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 16 * a.Cin * COUT; i += TPB) {
        int ci = i % a.Cin, co = (i / a.Cin) % COUT, tap = i / (COUT * a.Cin);
        wl[i] = a.w[(ci * COUT + co) * 16 + tap];
    }
    __syncthreads();
    const int OH = 2 * a.IH, OW = 2 * a.IW;
    const long long per_group = (long long)a.B * a.IH * a.IW;
    float s1[32], s2[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < per_group; i += (long long)gridDim.x * TPB) {
        const int ix = (int)(i % a.IW);
        long long t1 = i / a.IW;
        const int iy = (int)(t1 % a.IH);
        const long long n = (long long)g * a.B + t1 / a.IH;
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            const int oy = 2 * iy - 1 + kh;
            if ((unsigned)oy >= (unsigned)OH) continue;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int ox = 2 * ix - 1 + kw;
                if ((unsigned)ox >= (unsigned)OW) continue;
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float d = a.dlogit[((n * COUT + co) * OH + oy) * OW + ox];
                    const float* wp = wl + ((kh * 4 + kw) * COUT + co) * a.Cin;
#pragma unroll
                    for (int c = 0; c < 32; ++c) acc[c] += d * wp[c];
                }
            }
        }
        // d-activation of the producer layer (BatchNorm + act) and its two backward sums
        const long long pix = (n * a.IH + iy) * a.IW + ix;
        const bf16* rp = a.r + pix * a.Cin;
        bf16* op = a.db + pix * a.Cin;
#pragma unroll
        for (int c0 = 0; c0 < 32; c0 += 8) {
            bf16x8 rv = *reinterpret_cast<const bf16x8*>(rp + c0);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                const float rr = (float)rv[j];
                const float2 af = a.affine[g * a.Cin + c];
                const float2 mr = a.meanrstd[g * a.Cin + c];
                const float v = acc[c] * act_bwd(a.act, rr * af.x + af.y);
                s1[c] += v;
                s2[c] += v * (rr - mr.x) * mr.y;
                o[j] = (bf16)v;
            }
            *reinterpret_cast<bf16x8*>(op + c0) = o;
        }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        float x = wave_sum(s1[c]), y = wave_sum(s2[c]);
        if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6][c] = make_float2(x, y);
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float x = 0.f, y = 0.f;
        for (int w = 0; w < TPB / 64; ++w) { x += red_s[w][threadIdx.x].x; y += red_s[w][threadIdx.x].y; }
        atomicAdd(&a.red[g * a.Cin + threadIdx.x].x, x);
        atomicAdd(&a.red[g * a.Cin + threadIdx.x].y, y);
    }
}

}  // namespace

int launch_convt_last_fwd(const ConvTLastFwdArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.Cin == 32 && (a.Cout == 1 || a.Cout == 3) && a.G >= 1 && a.G <= 4, "convT last fwd: Cin=%d Cout=%d G=%d", a.Cin, a.Cout, a.G);
    long long per_group = (long long)a.B * 4 * a.IH * a.IW;
    dim3 grid((unsigned)min((per_group + TPB - 1) / TPB, (long long)4096), a.G);
    if (a.Cout == 1) hipLaunchKernelGGL(convt_last_fwd_kernel<1>, grid, dim3(TPB), 0, s, a);
    else hipLaunchKernelGGL(convt_last_fwd_kernel<3>, grid, dim3(TPB), 0, s, a);
    return mmvae_check_launch("convt_last_fwd");
}
int launch_convt_last_dgrad(const ConvTLastDgradArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.Cin == 32 && (a.Cout == 1 || a.Cout == 3) && a.G >= 1 && a.G <= 4, "convT last dgrad: Cin=%d Cout=%d", a.Cin, a.Cout);
    long long per_group = (long long)a.B * a.IH * a.IW;
    dim3 grid((unsigned)min((per_group + TPB - 1) / TPB, (long long)4096), a.G);
    if (a.Cout == 1) hipLaunchKernelGGL(convt_last_dgrad_kernel<1>, grid, dim3(TPB), 0, s, a);
    else hipLaunchKernelGGL(convt_last_dgrad_kernel<3>, grid, dim3(TPB), 0, s, a);
    return mmvae_check_launch("convt_last_dgrad");
}
