// Streaming kernels (HBM/L2 bound; 16-B vector accesses, wave reductions + one atomic per block).
#include "elementwise.h"
#include "bn_dev.h"
#include <math.h>

namespace {

constexpr int TPB = 256;
inline int nblocks(long long n, int per = TPB, int cap = 4096) {
    long long b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------ pack / unpack
// the descriptor whose block range holds `block` (first_block ascending, first_block[0] == 0).  One load per lane + a ballot per 64
// descriptors: the binary search this replaces was a chain of log2(nd) DEPENDENT global loads in front of every block's work
// (~3.5 us per block: the pack of the step prologue took 12 us for 20 MB, a launch that skipped most descriptors still 15 us).
__device__ __forceinline__ int find_desc(const PackDesc* t, int nd, int block) {
    const int lane = threadIdx.x & 63;
    int found = 0;
    for (int base = 0; base < nd; base += 64) {
        const int i = base + lane;
        const bool le = i < nd && t[i].first_block <= block;
        const unsigned long long m = __ballot(le);
        found += __popcll(m);
        if (m != ~0ull) break;
    }
    return __builtin_amdgcn_readfirstlane(found > 0 ? found - 1 : 0);
}
// q = r / d, rem = r % d for 0 <= r < 2^23 with inv = 1.0f/d (one correction step each way)
__device__ __forceinline__ int fast_divmod_ew(int r, int d, float inv, int& rem) {
    int q = (int)((float)r * inv);
    rem = r - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}
__device__ __forceinline__ long long pack_src(const PackDesc& d, int n, int k) {
    int nhi = n / d.NL, nlo = n - nhi * d.NL;
    int tap = k / d.C, c = k - tap * d.C;
    int ty = tap / d.TW, tx = tap - ty * d.TW;
    return d.src_off + (long long)nhi * d.s_nhi + (long long)nlo * d.s_nlo +
           (long long)(d.o_ty + ty * d.step_t) * d.s_ty + (long long)(d.o_tx + tx * d.step_t) * d.s_tx + (long long)c * d.s_c;
}

// one thread = 8 consecutive columns of one packed row (16-byte bf16 store); Kpad is a multiple of 8
__device__ __forceinline__ void pack_block(const PackDesc* __restrict__ table, int nd, const float* __restrict__ params,
                                           bf16* __restrict__ packed_bf, float* __restrict__ packed_f32, int block, unsigned parts = 0u) {
    const int di = find_desc(table, nd, block);
    const PackDesc d = table[di];
    if (parts != 0u && !((parts >> (d.part & 31)) & 1u)) return;
    // 32-bit index math with reciprocal divisions (packed matrices have < 2^23 vectors; launch_pack checks)
    const int v = (int)(block - d.first_block) * TPB + threadIdx.x;
    const int vpr = d.Kpad / 8;
    if (v >= d.Npad * vpr) return;
    int kv;
    const int n = fast_divmod_ew(v, vpr, 1.0f / (float)vpr, kv);
    const int k0 = kv * 8;
    float val[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) val[j] = 0.f;
    if (n < d.N) {
        int nlo;
        const int nhi = fast_divmod_ew(n, d.NL, 1.0f / (float)d.NL, nlo);
        const long long rowbase = d.src_off + (long long)nhi * d.s_nhi + (long long)nlo * d.s_nlo;
        if (d.C % 8 == 0 && k0 + 8 <= d.K) {              // the 8 columns share one tap
            int c, tx;
            const int tap = fast_divmod_ew(k0, d.C, 1.0f / (float)d.C, c);
            const int ty = fast_divmod_ew(tap, d.TW, 1.0f / (float)d.TW, tx);
            const long long b = rowbase + (long long)(d.o_ty + ty * d.step_t) * d.s_ty + (long long)(d.o_tx + tx * d.step_t) * d.s_tx + (long long)c * d.s_c;
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = params[b + (long long)j * d.s_c];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                if (k < d.K) val[j] = params[pack_src(d, n, k)];
                else if (k == d.K && d.bias_off >= 0) val[j] = params[d.bias_off + (long long)nhi * d.b_nhi + (long long)nlo * d.b_nlo];
            }
        }
    }
    const long long e = d.frag ? ((long long)((n >> 4) * (d.Kpad >> 5) + (kv >> 2)) * 64 + (kv & 3) * 16 + (n & 15)) * 8 : (long long)n * d.Kpad + k0;
    if (d.is_f32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) packed_f32[d.dst_off + e + j] = val[j];
    } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)val[j];
        *reinterpret_cast<bf16x8*>(packed_bf + d.dst_off + e) = o;
    }
}

// MAP = false: grads[flat] += packed value.  MAP = true: map[flat] = where that value lives (index into gmat, or
// -(index + 2) into gvec) -- built once per plan for the optimizer kernel that consumes the packed gradients directly.
__global__ __launch_bounds__(TPB) void pack_kernel(const PackDesc* __restrict__ table, int nd, const float* __restrict__ params,
                                                   bf16* __restrict__ packed_bf, float* __restrict__ packed_f32, int block0) {
    pack_block(table, nd, params, packed_bf, packed_f32, block0 + blockIdx.x);
}

template <bool MAP>
__global__ __launch_bounds__(TPB) void unpack_kernel(const PackDesc* __restrict__ table, int nd, const float* __restrict__ gmat,
                                                     const float* __restrict__ gvec, float* __restrict__ grads, int* __restrict__ map,
                                                     int part) {
    const int di = find_desc(table, nd, blockIdx.x);
    const PackDesc d = table[di];
    if (part >= 0 && d.part != part) return;
    const int v = (int)(blockIdx.x - d.first_block) * TPB + threadIdx.x;
    const int vpr = d.Kpad / 8;
    if (v >= d.N * vpr) return;
    int kv, nlo;
    const int n = fast_divmod_ew(v, vpr, 1.0f / (float)vpr, kv);
    const int k0 = kv * 8;
    if (k0 > d.K) return;
    const int nhi = fast_divmod_ew(n, d.NL, 1.0f / (float)d.NL, nlo);
    const long long sidx = d.dst_off + (long long)n * d.Kpad + k0;
    float val[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!MAP) {
        const float* src = (d.is_f32 ? gvec : gmat) + sidx;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { val[j] = lo[j]; val[4 + j] = hi[j]; }
    }
    auto put = [&](long long dst, int j) {
        if (MAP) map[dst] = d.is_f32 ? -(int)(sidx + j) - 2 : (int)(sidx + j);
        else grads[dst] += val[j];
    };
    if (d.C % 8 == 0 && k0 + 8 <= d.K) {
        const long long rowbase = d.src_off + (long long)nhi * d.s_nhi + (long long)nlo * d.s_nlo;
        int c, tx;
        const int tap = fast_divmod_ew(k0, d.C, 1.0f / (float)d.C, c);
        const int ty = fast_divmod_ew(tap, d.TW, 1.0f / (float)d.TW, tx);
        const long long b = rowbase + (long long)(d.o_ty + ty * d.step_t) * d.s_ty + (long long)(d.o_tx + tx * d.step_t) * d.s_tx + (long long)c * d.s_c;
#pragma unroll
        for (int j = 0; j < 8; ++j) put(b + (long long)j * d.s_c, j);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            if (k < d.K) put(pack_src(d, n, k), j);
            else if (k == d.K && d.bias_off >= 0) put(d.bias_off + (long long)nhi * d.b_nhi + (long long)nlo * d.b_nlo, j);
        }
    }
}

int table_blocks(const PackDesc* host, int nd) {
    const PackDesc& l = host[nd - 1];
    return l.first_block + (int)(((long long)l.Npad * (l.Kpad / 8) + TPB - 1) / TPB);
}

// ------------------------------------------------------------------ im2col for thin (1/3-channel) inputs
// one thread = one 8-wide vector of one patch row; 32-bit index math, reciprocal divisions, (tap, channel) walked
// incrementally over the 8 elements (the first version's 64-bit and per-element integer divisions made this
// streaming kernel 5x slower than its memory traffic)
__global__ __launch_bounds__(TPB) void im2col_small_kernel(const float* __restrict__ src, int Nimg, int Cin, int H, int W,
                                                           int KH, int KW, int stride, int pad, int OH, int OW,
                                                           bf16* __restrict__ dst, int ld) {
    const int vpr = ld / 8, rpb = TPB / vpr;                 // vectors per row, rows per block
    const int rl = threadIdx.x / vpr, kv = threadIdx.x - rl * vpr;
    if (rl >= rpb) return;
    const int nrows = Nimg * OH * OW;
    const int K = KH * KW * Cin, pix = OH * OW;
    const float inv_pix = 1.0f / (float)pix, inv_ow = 1.0f / (float)OW, inv_c = 1.0f / (float)Cin, inv_kw = 1.0f / (float)KW;
    for (int row = blockIdx.x * rpb + rl; row < nrows; row += gridDim.x * rpb) {
        int rem, ox;
        const int n = fast_divmod_ew(row, pix, inv_pix, rem);
        const int oy = fast_divmod_ew(rem, OW, inv_ow, ox);
        const int y0 = oy * stride - pad, x0 = ox * stride - pad;
        const int k0 = kv * 8;
        int ci, kw;
        int tap = fast_divmod_ew(k0, Cin, inv_c, ci);
        int kh = fast_divmod_ew(tap, KW, inv_kw, kw);
        const float* img = src + (size_t)n * Cin * H * W;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float val = 0.f;
            const int y = y0 + kh, x = x0 + kw;
            if (k0 + j < K && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) val = img[(ci * H + y) * W + x];
            o[j] = (bf16)val;
            if (++ci == Cin) { ci = 0; if (++kw == KW) { kw = 0; ++kh; } }
        }
        *reinterpret_cast<bf16x8*>(dst + (size_t)row * ld + k0) = o;
    }
}

// ------------------------------------------------------------------ BatchNorm finalize
__global__ void bn_finalize_kernel(const BnFinalizeArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    const float gamma = a.gamma[c], beta = a.beta[c];
    if (!a.training) {
        float rstd = rsqrtf(a.running_var[c] + a.eps);
        float mean = a.running_mean[c];
        for (int g = 0; g < a.G; ++g) {
            a.affine[g * a.C + c] = make_float2(gamma * rstd, beta - mean * gamma * rstd);
            a.meanrstd[g * a.C + c] = make_float2(mean, rstd);
        }
        return;
    }
    float rm = a.running_mean ? a.running_mean[c] : 0.f;
    float rv = a.running_var ? a.running_var[c] : 0.f;
    for (int g = 0; g < a.G; ++g) {
        float2 s = make_float2(0.f, 0.f);
        for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
            float2 t = a.stats[(g * MMVAE_STAT_SLOTS + q) * a.C + c];
            s.x += t.x; s.y += t.y;
        }
        float mean = s.x / a.count;
        float var = fmaxf(s.y / a.count - mean * mean, 0.f);
        float rstd = rsqrtf(var + a.eps);
        a.affine[g * a.C + c] = make_float2(gamma * rstd, beta - mean * gamma * rstd);
        a.meanrstd[g * a.C + c] = make_float2(mean, rstd);
        float unbiased = var * a.count / (a.count - 1.f);
        const int nu = ((a.skip_update_mask >> g) & 1u) ? 0 : a.updates_per_group;
        for (int u = 0; u < nu; ++u) {
            rm = (1.f - a.momentum) * rm + a.momentum * mean;
            rv = (1.f - a.momentum) * rv + a.momentum * unbiased;
        }
    }
    if (a.running_mean) { a.running_mean[c] = rm; a.running_var[c] = rv; }
    if (c == 0 && a.num_batches_tracked)
        *a.num_batches_tracked += (long long)(a.G - __popc(a.skip_update_mask & ((1u << a.G) - 1u))) * a.updates_per_group;
}

// ------------------------------------------------------------------ BatchNorm finalize + normalise + activation
__global__ __launch_bounds__(TPB) void bn_act_kernel(const BnActArgs a) {
    extern __shared__ float2 aff_s[];        // [G][C]
    const BnFinalizeArgs& f = a.fin;
    for (int i = threadIdx.x; i < a.G * a.C; i += TPB) {
        float2 aff, mr;
        bn_channel_tables(f, i / a.C, i % a.C, aff, mr);
        aff_s[i] = aff;
        if (blockIdx.x == 0 && f.affine) { f.affine[i] = aff; f.meanrstd[i] = mr; }
    }
    if (blockIdx.x == 0) bn_running_update(f, threadIdx.x, TPB);
    __syncthreads();
    const int vpr = (a.C + 7) / 8;           // C need not be a multiple of 8: pad columns (< ld) are written as zero
    const long long nvec = (long long)a.rows * vpr;
    for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < nvec; v += (long long)gridDim.x * TPB) {
        const long long row = v / vpr;
        const int c0 = (int)(v - row * vpr) * 8;
        const int g = (int)(row / a.rows_per_group);
        bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.r + row * a.ld + c0);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float val = 0.f;
            if (c0 + j < a.C) {
                float2 af = aff_s[g * a.C + c0 + j];
                val = act_fwd(a.act, (float)rv[j] * af.x + af.y);
            }
            o[j] = (bf16)val;
        }
        *reinterpret_cast<bf16x8*>(a.a + row * a.ld + c0) = o;
    }
}

// ------------------------------------------------------------------ BatchNorm backward apply
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const BnBwdApplyArgs a) {
    extern __shared__ float4 tab_s[];        // [G][C]: (sum db / n, sum db*xhat / n, mean, gamma*rstd)
    const float inv_cnt = 1.f / (float)a.rows_per_group;
    for (int i = threadIdx.x; i < a.G * a.C; i += TPB) {
        const int g = i / a.C, c = i - g * a.C;
        float sx = 0.f, sy = 0.f;
        for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
            float2 t = a.red[(g * MMVAE_STAT_SLOTS + q) * a.C + c];
            sx += t.x; sy += t.y;
        }
        const float2 mr = a.meanrstd[i];
        tab_s[i] = make_float4(sx * inv_cnt, sy * inv_cnt, mr.x, mr.y);
        if (blockIdx.x == 0 && g == 0) {       // parameter gradients: summed over groups by the threads of group 0
            float tg = 0.f, tb = 0.f;
            for (int gg = 0; gg < a.G; ++gg)
                for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
                    float2 t = a.red[(gg * MMVAE_STAT_SLOTS + q) * a.C + c];
                    tb += t.x; tg += t.y;
                }
            if (a.dgamma) a.dgamma[c] += tg;
            if (a.dbeta) a.dbeta[c] += tb;
        }
    }
    __syncthreads();
    const int vpr = (a.C + 7) / 8;
    const long long nvec = (long long)a.rows * vpr;
    for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < nvec; v += (long long)gridDim.x * TPB) {
        const long long row = v / vpr;
        const int c0 = (int)(v - row * vpr) * 8;
        const int g = (int)(row / a.rows_per_group);
        bf16x8 dbv = *reinterpret_cast<const bf16x8*>(a.db + row * a.ld + c0);
        bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.r + row * a.ld + c0);
        float dbf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dbf[j] = (float)dbv[j];
        if (a.db2) {
            bf16x8 d2 = *reinterpret_cast<const bf16x8*>(a.db2 + row * a.ld + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) dbf[j] += (float)d2[j];
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            float val = 0.f;
            if (c < a.C) {
                const float4 t = tab_s[g * a.C + c];
                const float xh = ((float)rv[j] - t.z) * t.w;
                val = a.gamma[c] * t.w * (dbf[j] - t.x - xh * t.y);
            }
            o[j] = (bf16)val;
        }
        *reinterpret_cast<bf16x8*>(a.dr + row * a.ld + c0) = o;
    }
}

// ------------------------------------------------------------------ sigmoid + binary cross entropy
__global__ __launch_bounds__(TPB) void sigmoid_bce_kernel(const BceArgs a) {
    const int g = blockIdx.y;
    const long long per_group = (long long)a.B * a.C * a.H * a.W;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < per_group; i += (long long)gridDim.x * TPB) {
        // i indexes NCHW of the target
        const int x = (int)(i % a.W);
        long long t1 = i / a.W;
        const int y = (int)(t1 % a.H);
        t1 /= a.H;
        const int c = (int)(t1 % a.C);
        const int n = (int)(t1 / a.C);
        const long long ng = (long long)g * a.B + n;
        const float l = a.logits[((ng * a.H + y) * a.W + x) * a.ldl + c];
        const float t = a.target[i];
        const float p = 1.0f / (1.0f + expf(-l));                  // F.sigmoid
        const float lp = fmaxf(logf(p), -100.f);                    // F.binary_cross_entropy clamps both logs at -100
        const float lq = fmaxf(logf(1.0f - p), -100.f);
        acc += -(t * lp + (1.0f - t) * lq);
        const long long o = g * per_group + i;
        if (a.recon) a.recon[o] = p;
        if (a.dlogit) {
            const float pq = p * (1.0f - p);
            a.dlogit[o] = a.coef[g] * (p - t) / fmaxf(pq, 1e-12f) * pq;   // BCE backward (EPS 1e-12) x sigmoid backward
        }
    }
    acc = wave_sum(acc);
    __shared__ float part[TPB / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < TPB / 64; ++w) s += part[w];
        atomicAdd(a.loss_sum + g, s);
    }
}

// ------------------------------------------------------------------ product of experts / reparam / KL
struct Poe2 { float mu, lv; };
// forward of multimnist/model.py:355-360 for M experts held in registers
template <int M>
__device__ __forceinline__ Poe2 poe_fwd_m(const float (&mu)[M], const float (&lv)[M]) {
    float s0 = 0.f, s1 = 0.f, t = 0.f;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        float var = expf(lv[i]) + 1e-8f;
        s0 += var; s1 += mu[i] * var; t += 1.0f / var;
    }
    Poe2 o;
    o.mu = s1 / s0;
    o.lv = logf(1.0f / t);
    return o;
}
template <int M>
__device__ __forceinline__ void poe_bwd_m(const float (&mu)[M], const float (&lv)[M], float gmu, float glv,
                                          float (&dmu)[M], float (&dlv)[M]) {
    float s0 = 0.f, s1 = 0.f, t = 0.f;
    float var[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        var[i] = expf(lv[i]) + 1e-8f;
        s0 += var[i]; s1 += mu[i] * var[i]; t += 1.0f / var[i];
    }
    const float pm = s1 / s0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        dmu[i] = gmu * var[i] / s0;
        float dvar = gmu * (mu[i] - pm) / s0 + glv / (t * var[i] * var[i]);
        dlv[i] = dvar * expf(lv[i]);
    }
}

__global__ __launch_bounds__(TPB) void poe_fwd_kernel(const float* mu, const float* lv, int M, int n, float* omu, float* olv) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, t = 0.f;
    for (int m = 0; m < M; ++m) {
        float var = expf(lv[(long long)m * n + i]) + 1e-8f;
        s0 += var; s1 += mu[(long long)m * n + i] * var; t += 1.0f / var;
    }
    omu[i] = s1 / s0;
    olv[i] = logf(1.0f / t);
}
__global__ __launch_bounds__(TPB) void poe_bwd_kernel(const float* mu, const float* lv, int M, int n, const float* gmu,
                                                      const float* glv, float* dmu, float* dlv) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, t = 0.f;
    for (int m = 0; m < M; ++m) {
        float var = expf(lv[(long long)m * n + i]) + 1e-8f;
        s0 += var; s1 += mu[(long long)m * n + i] * var; t += 1.0f / var;
    }
    const float pm = s1 / s0, a = gmu[i], b = glv[i];
    for (int m = 0; m < M; ++m) {
        float e = expf(lv[(long long)m * n + i]);
        float var = e + 1e-8f;
        dmu[(long long)m * n + i] = a * var / s0;
        dlv[(long long)m * n + i] = (a * (mu[(long long)m * n + i] - pm) / s0 + b / (t * var * var)) * e;
    }
}
__global__ __launch_bounds__(TPB) void reparam_fwd_kernel(const float* mu, const float* lv, const float* eps, int n, float* z) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n) z[i] = eps[i] * expf(0.5f * lv[i]) + mu[i];
}
__global__ __launch_bounds__(TPB) void reparam_bwd_kernel(const float* lv, const float* eps, const float* dz, int n, float* dmu, float* dlv) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n) { dmu[i] = dz[i]; dlv[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]); }
}
__global__ __launch_bounds__(TPB) void kl_fwd_kernel(const float* mu, const float* lv, int n, float* out) {
    float acc = 0.f;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB)
        acc += -0.5f * (1.0f + lv[i] - mu[i] * mu[i] - expf(lv[i]));
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(TPB) void kl_bwd_kernel(const float* mu, const float* lv, int n, float coef, const float* gs, float* dmu, float* dlv) {
    if (gs) coef *= *gs;                                   // upstream gradient (0-d device tensor): no host read
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n) { dmu[i] = coef * mu[i]; dlv[i] = -0.5f * coef * (1.0f - expf(lv[i])); }
}

__global__ __launch_bounds__(TPB) void normal_kernel(float* out, long long n, unsigned long long seed, const long long* step, unsigned stream_id) {
    const unsigned long long st = step ? (unsigned long long)*step : 0ull;
    for (long long q = (long long)blockIdx.x * TPB + threadIdx.x; q * 4 < n; q += (long long)gridDim.x * TPB) {
        uint32_t r[4];
        Philox::gen(seed ^ (st * 0x9E3779B97F4A7C15ull), (uint64_t)q, stream_id, r);
        float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
        float m0 = sqrtf(-2.0f * logf(u0)), m1 = sqrtf(-2.0f * logf(u2));
        float v[4] = {m0 * cosf(6.28318530718f * u1), m0 * sinf(6.28318530718f * u1),
                      m1 * cosf(6.28318530718f * u3), m1 * sinf(6.28318530718f * u3)};
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = v[j];
    }
}
__global__ __launch_bounds__(TPB) void keep_mask_kernel(uint8_t* out, long long n, float p, unsigned long long seed, const long long* step, unsigned stream_id) {
    const unsigned long long st = step ? (unsigned long long)*step : 0ull;
    for (long long q = (long long)blockIdx.x * TPB + threadIdx.x; q * 4 < n; q += (long long)gridDim.x * TPB) {
        uint32_t r[4];
        Philox::gen(seed ^ (st * 0x9E3779B97F4A7C15ull), (uint64_t)q, stream_id, r);
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = u01(r[j]) >= p ? 1 : 0;
    }
}

// ------------------------------------------------------------------ step prologue: zeroing + RNG in one launch
__global__ __launch_bounds__(TPB) void step_begin_kernel(const StepBeginArgs a) {
    // the first pack_blocks workgroups refresh the bf16 GEMM copies of the weights (what pack_kernel does: after an
    // optimizer step); the others zero the accumulators and draw the step's random numbers -- independent work, one launch
    if ((int)blockIdx.x < a.pack_blocks) {
        pack_block(a.pack_table, a.pack_nd, a.pack_params, a.packed_bf, a.packed_f32, blockIdx.x, a.pack_parts);
        return;
    }
    const long long gtid = (long long)(blockIdx.x - a.pack_blocks) * TPB + threadIdx.x, gstride = (long long)(gridDim.x - a.pack_blocks) * TPB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (!a.zero_ptr[r]) continue;
        u32x4* p = reinterpret_cast<u32x4*>(a.zero_ptr[r]);
        const long long n16 = (long long)(a.zero_bytes[r] / 16);
        for (long long i = gtid; i < n16; i += gstride) p[i] = u32x4{0u, 0u, 0u, 0u};
    }
    const unsigned long long st = a.step ? (unsigned long long)*a.step : 0ull;
    const unsigned long long key = a.seed ^ (st * 0x9E3779B97F4A7C15ull);
    if (a.eps) {
        for (long long q = gtid; q * 4 < a.n_eps; q += gstride) {
            uint32_t r[4];
            Philox::gen(key, (uint64_t)q, 1, r);
            float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
            float m0 = sqrtf(-2.0f * logf(u0)), m1 = sqrtf(-2.0f * logf(u2));
            float v[4] = {m0 * cosf(6.28318530718f * u1), m0 * sinf(6.28318530718f * u1),
                          m1 * cosf(6.28318530718f * u3), m1 * sinf(6.28318530718f * u3)};
            for (int j = 0; j < 4; ++j)
                if (q * 4 + j < a.n_eps) a.eps[q * 4 + j] = v[j];
        }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (!a.mask[m]) continue;
        for (long long q = gtid; q * 4 < a.n_mask[m]; q += gstride) {
            uint32_t r[4];
            Philox::gen(key, (uint64_t)q, 2 + m, r);
            for (int j = 0; j < 4; ++j)
                if (q * 4 + j < a.n_mask[m]) a.mask[m][q * 4 + j] = u01(r[j]) >= a.p ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------ standalone loss pieces (drop-in loss_function)
__global__ __launch_bounds__(TPB) void bce_fwd_kernel(const float* p, const float* t, long long n, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        const float lp = fmaxf(logf(p[i]), -100.f), lq = fmaxf(logf(1.0f - p[i]), -100.f);
        acc += -(t[i] * lp + (1.0f - t[i]) * lq);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(TPB) void bce_bwd_kernel(const float* p, const float* t, long long n, float coef, const float* gs, float* dp) {
    if (gs) coef *= *gs;
    long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) dp[i] = coef * (p[i] - t[i]) / fmaxf((1.0f - p[i]) * p[i], 1e-12f);
}
__global__ __launch_bounds__(TPB) void nll_fwd_kernel(const float* lp, const long long* tg, int rows, int classes, float* out) {
    float acc = 0.f;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < rows; i += gridDim.x * TPB) acc += -lp[(size_t)i * classes + tg[i]];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(TPB) void nll_bwd_kernel(const long long* tg, int rows, int classes, float coef, const float* gs, float* dlp) {
    if (gs) coef *= *gs;
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i < rows * classes) dlp[i] = (i % classes == (int)tg[i / classes]) ? -coef : 0.f;
}

// ------------------------------------------------------------------ fused 3-pass latent block
// pass 0: experts {image(pass-1 encoder output), text}; pass 1: {image(pass-2 output)}; pass 2: {text}
__global__ __launch_bounds__(TPB) void latent3_fwd_kernel(const Latent3Args a) {
    const int n = a.B * a.D;
    float kl[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < a.B * a.ldz; i += gridDim.x * TPB) {
        const int b = i / a.ldz, d = i - b * a.ldz;
        if (d >= a.D) {        // column D carries the 1.0 that multiplies the folded bias; the other pads are zero
            for (int k = 0; k < 3; ++k) a.z_bf[(size_t)(k * a.B + b) * a.ldz + d] = (bf16)(d == a.D ? 1.f : 0.f);
            continue;
        }
        const int D2 = 2 * a.D;
        const float im0 = a.img_out[(size_t)b * D2 + d], il0 = a.img_out[(size_t)b * D2 + a.D + d];
        const float* img_b = a.img_out_b ? a.img_out_b : a.img_out + (size_t)a.B * D2;
        const float im1 = img_b[(size_t)b * D2 + d], il1 = img_b[(size_t)b * D2 + a.D + d];
        const float tm = a.txt_out[(size_t)b * D2 + d], tl = a.txt_out[(size_t)b * D2 + a.D + d];
        Poe2 o[3];
        { float m[2] = {im0, tm}, l[2] = {il0, tl}; o[0] = poe_fwd_m<2>(m, l); }
        { float m[1] = {im1}, l[1] = {il1}; o[1] = poe_fwd_m<1>(m, l); }
        { float m[1] = {tm}, l[1] = {tl}; o[2] = poe_fwd_m<1>(m, l); }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t e = (size_t)k * n + (size_t)b * a.D + d;
            a.mu[e] = o[k].mu; a.logvar[e] = o[k].lv;
            float z = a.training ? a.eps[e] * expf(0.5f * o[k].lv) + o[k].mu : o[k].mu;
            a.z_f32[e] = z;
            a.z_bf[(size_t)(k * a.B + b) * a.ldz + d] = (bf16)z;
            kl[k] += -0.5f * (1.0f + o[k].lv - o[k].mu * o[k].mu - expf(o[k].lv));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = wave_sum(kl[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.kl_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + k, s);
    }
}

// Thread (tx, ty) of a 64 x 4 block: latent column d = 64*blockIdx.x + tx, sample rows b = L3B_ROWS*blockIdx.y + ty + 4*i.
// The bias gradients (column sums over the batch) are reduced in registers and through LDS: ONE atomic per column per block
// (the first version issued 4 atomics per element onto 2D addresses: same-address float atomics serialise at the memory
// side, 16 us for a 25k-element kernel on the main chain).  Block (0,0) also folds the step's loss slots into io.sums.
constexpr int L3B_ROWS = 4;     // one row per thread: B/4 x ceil(D/64) workgroups (the kernel is a latency chain on the main stream)
__global__ __launch_bounds__(TPB) void latent3_bwd_kernel(const Latent3BwdArgs a) {
    const Latent3Args& f = a.f;
    const int n = f.B * f.D, D2 = 2 * f.D;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + tx;
    __shared__ float red[4][4][64];
    if (a.loss_slots && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) {
        float t = 0.f;
        for (int q = 0; q < MMVAE_LOSS_SLOTS; ++q) t += a.loss_slots[q * 16 + threadIdx.x];
        a.loss_out[threadIdx.x] = t;
    }
    float s_im = 0.f, s_il = 0.f, s_tm = 0.f, s_tl = 0.f;
    const int b_end = min(f.B, (int)(blockIdx.y + 1) * L3B_ROWS);
    if (d < f.D)
    for (int b = blockIdx.y * L3B_ROWS + ty; b < b_end; b += 4) {
        const int i = b * f.D + d;
        const float im0 = f.img_out[(size_t)b * D2 + d], il0 = f.img_out[(size_t)b * D2 + f.D + d];
        const float* img_b = f.img_out_b ? f.img_out_b : f.img_out + (size_t)f.B * D2;
        const float im1 = img_b[(size_t)b * D2 + d], il1 = img_b[(size_t)b * D2 + f.D + d];
        const float tm = f.txt_out[(size_t)b * D2 + d], tl = f.txt_out[(size_t)b * D2 + f.D + d];
        float gmu[3], glv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t e = (size_t)k * n + i;
            float dz = 0.f;
            if (a.dz_a) dz += a.dz_a[e];
            if (a.dz_b) dz += a.dz_b[e];
            const float mu = f.mu[e], lv = f.logvar[e];
            gmu[k] = dz + a.kl_coef[k] * mu;
            glv[k] = -0.5f * a.kl_coef[k] * (1.0f - expf(lv));
            if (f.training) glv[k] += dz * f.eps[e] * 0.5f * expf(0.5f * lv);
        }
        float d_im0, d_il0, d_im1, d_il1, d_tm = 0.f, d_tl = 0.f;
        {
            float m[2] = {im0, tm}, l[2] = {il0, tl}, dm[2], dl[2];
            poe_bwd_m<2>(m, l, gmu[0], glv[0], dm, dl);
            d_im0 = dm[0]; d_il0 = dl[0]; d_tm += dm[1]; d_tl += dl[1];
        }
        {
            float m[1] = {im1}, l[1] = {il1}, dm[1], dl[1];
            poe_bwd_m<1>(m, l, gmu[1], glv[1], dm, dl);
            d_im1 = dm[0]; d_il1 = dl[0];
        }
        {
            float m[1] = {tm}, l[1] = {tl}, dm[1], dl[1];
            poe_bwd_m<1>(m, l, gmu[2], glv[2], dm, dl);
            d_tm += dm[0]; d_tl += dl[0];
        }
        if (a.d_img_out_f32) {          // fp32 model path (MNIST): summed gradient of the two passes that share the encoder
            a.d_img_out_f32[(size_t)b * D2 + d] = d_im0 + d_im1;
            a.d_img_out_f32[(size_t)b * D2 + f.D + d] = d_il0 + d_il1;
        } else if (a.sum_img_variants) {       // the two variants share one encoder forward: its backward needs the summed gradient
            a.d_img_out_bf[(size_t)b * D2 + d] = (bf16)(d_im0 + d_im1);
            a.d_img_out_bf[(size_t)b * D2 + f.D + d] = (bf16)(d_il0 + d_il1);
        } else {
            a.d_img_out_bf[(size_t)b * D2 + d] = (bf16)d_im0;
            a.d_img_out_bf[(size_t)b * D2 + f.D + d] = (bf16)d_il0;
            a.d_img_out_bf[(size_t)(f.B + b) * D2 + d] = (bf16)d_im1;
            a.d_img_out_bf[(size_t)(f.B + b) * D2 + f.D + d] = (bf16)d_il1;
        }
        if (a.d_txt_out) {
            a.d_txt_out[(size_t)b * D2 + d] = d_tm;
            a.d_txt_out[(size_t)b * D2 + f.D + d] = d_tl;
        }
        if (a.d_txt_out_bf) {
            a.d_txt_out_bf[(size_t)b * D2 + d] = (bf16)d_tm;
            a.d_txt_out_bf[(size_t)b * D2 + f.D + d] = (bf16)d_tl;
        }
        s_im += d_im0 + d_im1; s_il += d_il0 + d_il1; s_tm += d_tm; s_tl += d_tl;
    }
    red[0][ty][tx] = s_im; red[1][ty][tx] = s_il; red[2][ty][tx] = s_tm; red[3][ty][tx] = s_tl;
    __syncthreads();
    if (ty == 0 && d < f.D) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = red[q][0][tx] + red[q][1][tx] + red[q][2][tx] + red[q][3][tx];
        if (a.d_img_bias) { atomicAdd(a.d_img_bias + d, v[0]); atomicAdd(a.d_img_bias + f.D + d, v[1]); }
        if (a.d_txt_bias) { atomicAdd(a.d_txt_bias + d, v[2]); atomicAdd(a.d_txt_bias + f.D + d, v[3]); }
    }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults)
__global__ __launch_bounds__(TPB) void adam_kernel(const AdamArgs a, unsigned* done) {
    const long long t = *a.step + 1;
    // a step that gave up on a device-side exchange marked itself (plan_base.h sum_slots_kernel): no update, no step count
    // (the mark is a NaN with a payload of its own, compared bit by bit without the sign: any other NaN is a real gradient value)
    const bool skip = done[1] != 0u || (__float_as_uint(a.g[0]) & 0x7fffffffu) == MMVAE_VOID_MARK;
    const float bc1 = 1.0f - powf(a.b1, (float)t);
    const float bc2 = 1.0f - powf(a.b2, (float)t);
    const float step_size = a.lr / bc1;
    const float inv_sqrt_bc2 = rsqrtf(bc2);
    // element ranges of this launch (launch_adam: one range [0, n) unless the caller gave its own), walked as one list of 4-element vectors
    long long vbase[5];
    vbase[0] = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) vbase[r + 1] = vbase[r] + (r < a.nr ? (a.rlen[r] + 3) / 4 : 0);
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < vbase[4] && !skip; i += (long long)gridDim.x * TPB) {
        int r = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) r += (q < a.nr && i >= vbase[q]) ? 1 : 0;
        const long long e = a.roff[r] + (i - vbase[r]) * 4, r_end = a.roff[r] + a.rlen[r];
        if (e + 4 <= r_end) {
            f32x4 g = *reinterpret_cast<const f32x4*>(a.g + e);
            if (a.gmap) {       // gradient of the GEMM weights still in its packed layout: gather, add, keep the flat copy
                const int4 mi = *reinterpret_cast<const int4*>(a.gmap + e);
                const int mm[4] = {mi.x, mi.y, mi.z, mi.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (mm[j] >= 0) g[j] += a.gpk[mm[j]];
                    else if (mm[j] < -1) g[j] += a.gpk_vec[-mm[j] - 2];
                *reinterpret_cast<f32x4*>(a.g_out + e) = g;
            }
            f32x4 m = *reinterpret_cast<const f32x4*>(a.m + e);
            f32x4 v = *reinterpret_cast<const f32x4*>(a.v + e);
            f32x4 p = *reinterpret_cast<const f32x4*>(a.p + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gj = g[j] * a.grad_scale;
                m[j] = a.b1 * m[j] + (1.f - a.b1) * gj;
                v[j] = a.b2 * v[j] + (1.f - a.b2) * gj * gj;
                p[j] -= step_size * m[j] / (sqrtf(v[j]) * inv_sqrt_bc2 + a.eps);
            }
            *reinterpret_cast<f32x4*>(a.m + e) = m;
            *reinterpret_cast<f32x4*>(a.v + e) = v;
            *reinterpret_cast<f32x4*>(a.p + e) = p;
        } else {
            for (long long q = e; q < r_end; ++q) {
                float gq = a.g[q];
                if (a.gmap) {
                    const int mq = a.gmap[q];
                    if (mq >= 0) gq += a.gpk[mq]; else if (mq < -1) gq += a.gpk_vec[-mq - 2];
                    a.g_out[q] = gq;
                }
                float gj = gq * a.grad_scale;
                float m = a.b1 * a.m[q] + (1.f - a.b1) * gj;
                float v = a.b2 * a.v[q] + (1.f - a.b2) * gj * gj;
                a.m[q] = m; a.v[q] = v;
                a.p[q] -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
            }
        }
    }
    // the last block to finish advances the step counter (every block has read it by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned ticket = atomicAdd(done, 1u);
        if (ticket == gridDim.x - 1) {
            done[0] = 0u;
            done[1] = 0u;
            if (!skip && !a.no_advance) *a.step = t;
            __threadfence();
        }
    }
}

__global__ __launch_bounds__(TPB) void embed_gather_stats_kernel(const float* table, int C, const long long* idx, int rows, int idx_rows,
                                                                 int rows_per_group, bf16* x, int ld, float2* colstats) {
    // block = 64 rows x all channels (thread t: channel t % 64 .., rows strided)
    const int r0 = blockIdx.x * 64;
    for (int c = threadIdx.x & 63; c < ld; c += 64) {
        float s1 = 0.f, s2 = 0.f;
        int g_cur = -1;
        for (int rr = threadIdx.x >> 6; rr < 64; rr += TPB / 64) {
            const int r = r0 + rr;
            if (r >= rows) break;
            const int g = r / rows_per_group;
            if (g != g_cur && g_cur >= 0 && c < C && colstats) {
                atomicAdd(&colstats[(g_cur * MMVAE_STAT_SLOTS + blockIdx.x % MMVAE_STAT_SLOTS) * C + c].x, s1);
                atomicAdd(&colstats[(g_cur * MMVAE_STAT_SLOTS + blockIdx.x % MMVAE_STAT_SLOTS) * C + c].y, s2);
                s1 = s2 = 0.f;
            }
            g_cur = g;
            float v = 0.f;
            if (c < C) v = table[idx[r % idx_rows] * C + c];
            const bf16 b = (bf16)v;
            x[(size_t)r * ld + c] = b;
            const float vb = (float)b;
            s1 += vb; s2 += vb * vb;
        }
        if (g_cur >= 0 && c < C && colstats) {
            atomicAdd(&colstats[(g_cur * MMVAE_STAT_SLOTS + blockIdx.x % MMVAE_STAT_SLOTS) * C + c].x, s1);
            atomicAdd(&colstats[(g_cur * MMVAE_STAT_SLOTS + blockIdx.x % MMVAE_STAT_SLOTS) * C + c].y, s2);
        }
    }
}
__global__ __launch_bounds__(TPB) void embed_scatter_add_kernel(const bf16* d, int ld, int C, const long long* idx, int rows, float* g_table) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= rows * C) return;
    const int r = i / C, c = i - r * C;
    atomicAdd(g_table + idx[r] * C + c, (float)d[(size_t)r * ld + c]);
}
__global__ __launch_bounds__(TPB) void logsoftmax_nll_kernel(const LogSoftmaxNllArgs a) {
    const int r = blockIdx.x * TPB + threadIdx.x;
    float nll = 0.f;
    int g = 0;
    if (r < a.rows) {
        g = r / a.rows_per_group;
        const float* l = a.logits + (size_t)r * a.classes;
        float mx = -INFINITY;
        for (int c = 0; c < a.classes; ++c) mx = fmaxf(mx, l[c]);
        float se = 0.f;
        for (int c = 0; c < a.classes; ++c) se += expf(l[c] - mx);
        const float lse = mx + logf(se);
        const int tg = a.target ? (int)a.target[r % a.target_rows] : -1;
        for (int c = 0; c < a.classes; ++c) {
            const float lp = l[c] - lse;
            a.words[(size_t)r * a.classes + c] = lp;
            if (c == tg) nll = -lp;
            if (a.dlogits) a.dlogits[(size_t)r * a.ld_d + c] = (bf16)(a.coef[g & 3] * (expf(lp) - (c == tg ? 1.f : 0.f)));
            if (a.dlogits_f32) a.dlogits_f32[(size_t)r * a.classes + c] = a.coef[g & 3] * (expf(lp) - (c == tg ? 1.f : 0.f));
        }
        if (a.dlogits)
            for (int c = a.classes; c < a.ld_d; ++c) a.dlogits[(size_t)r * a.ld_d + c] = (bf16)0.f;
    }
    if (a.nll_sum && a.target) {
        // rows of one wave may straddle a group boundary only when rows_per_group % 64 != 0: add per lane then
        if (a.rows_per_group % 64 == 0) {
            float s = wave_sum(nll);
            if ((threadIdx.x & 63) == 0 && r < a.rows) atomicAdd(a.nll_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + g, s);
        } else if (r < a.rows) {
            atomicAdd(a.nll_sum + (r % MMVAE_LOSS_SLOTS) * 16 + g, nll);
        }
    }
}
__global__ __launch_bounds__(TPB) void cast_bf16_kernel(const float* x, long long n, bf16* out) {
    const long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        bf16x4 o;
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x4*>(out + i) = o;
    } else {
        for (long long q = i; q < n; ++q) out[q] = (bf16)x[q];
    }
}
// F.mse_loss pieces (coco/train.py:75): out += sum (a-b)^2 ; d_a = coef * 2 (a-b)
__global__ __launch_bounds__(TPB) void mse_fwd_kernel(const float* a, const float* b, long long n, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        const float d = a[i] - b[i];
        acc += d * d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(TPB) void mse_bwd_kernel(const float* a, const float* b, long long n, float coef, const float* gs, float* da) {
    if (gs) coef *= *gs;
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) da[i] = coef * 2.f * (a[i] - b[i]);
}
// torchvision ToTensor on the device: uint8 pixels -> fp32 / denom (255; a true division, bit-equal to .div(255)), 16 pixels per thread
__global__ __launch_bounds__(TPB) void u8_to_f32_kernel(const uint8_t* __restrict__ src, long long n, float denom, float* __restrict__ dst) {
    const long long i = ((long long)blockIdx.x * TPB + threadIdx.x) * 16;
    if (i + 16 <= n && (reinterpret_cast<uintptr_t>(src + i) & 15) == 0) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = __fdiv_rn((float)((v[q] >> (8 * j)) & 0xffu), denom);
            *reinterpret_cast<f32x4*>(dst + i + 4 * q) = o;
        }
    } else {
        for (long long q = i; q < n && q < i + 16; ++q) dst[q] = __fdiv_rn((float)src[q], denom);
    }
}
template <typename T>
__global__ __launch_bounds__(TPB) void colsum_kernel_t(const T* x, int ld, int rows, int cols, float* out) {
    // block = 64 columns x (TPB/64) row lanes; partial sums through LDS
    __shared__ float part[TPB / 64][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    float s = 0.f;
    if (c < cols)
        for (int r = blockIdx.y * (TPB / 64) + (threadIdx.x >> 6); r < rows; r += gridDim.y * (TPB / 64)) s += (float)x[(size_t)r * ld + c];
    part[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && c < cols) {
        float t = 0.f;
        for (int w = 0; w < TPB / 64; ++w) t += part[w][threadIdx.x];
        atomicAdd(out + c, t);
    }
}

// bf16 rows of ld (multiple of 8) elements: a thread owns one 8-column vector (16-byte loads) of every RL-th row
__global__ __launch_bounds__(TPB) void colsum_bf16_kernel(const bf16* x, int ld, int rows, int cols, float* out) {
    __shared__ float part[TPB][8];
    const int vpr = ld >> 3, RL = TPB / vpr;
    const int rl = threadIdx.x / vpr, v = threadIdx.x - rl * vpr;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
        const long long step = (long long)gridDim.x * RL;
        long long r = (long long)blockIdx.x * RL + rl;
        for (; r + 3 * step < rows; r += 4 * step) {        // four rows in flight per thread
            bf16x8 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const bf16x8*>(x + (size_t)(r + u * step) * ld + v * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)q[u][j];
        }
        for (; r < rows; r += step) {
            const bf16x8 q = *reinterpret_cast<const bf16x8*>(x + (size_t)r * ld + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)q[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[threadIdx.x][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < vpr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = 0.f;
            for (int w = 0; w < RL; ++w) t += part[w * vpr + threadIdx.x][j];
            if (threadIdx.x * 8 + j < cols) atomicAdd(out + threadIdx.x * 8 + j, t);
        }
    }
}

}  // namespace

int launch_mse_fwd(const float* a, const float* b, long long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(nblocks(n, TPB * 4, 256)), dim3(TPB), 0, s, a, b, n, out);
    return mmvae_check_launch("mse_fwd");
}
int launch_mse_bwd(const float* a, const float* b, long long n, float coef, const float* gs, float* da, hipStream_t s) {
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, a, b, n, coef, gs, da);
    return mmvae_check_launch("mse_bwd");
}
int launch_u8_to_f32(const uint8_t* src, long long n, float denom, float* dst, hipStream_t s) {
    MMVAE_REQUIRE(src && dst && n >= 0, "u8_to_f32: null argument");
    if (n == 0) return MMVAE_OK;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)((n + 16 * TPB - 1) / (16 * TPB))), dim3(TPB), 0, s, src, n, denom, dst);
    return mmvae_check_launch("u8_to_f32");
}
int launch_pack(const PackDesc* table_dev, const PackDesc* table_host, int nd, const float* params, bf16* packed_bf,
                float* packed_f32, hipStream_t s) {
    MMVAE_REQUIRE(nd > 0, "pack: empty table");
    for (int i = 0; i < nd; ++i)
        MMVAE_REQUIRE((long long)table_host[i].Npad * (table_host[i].Kpad / 8) < (1 << 23), "pack: matrix %d too large for 32-bit index math", i);
    hipLaunchKernelGGL(pack_kernel, dim3(table_blocks(table_host, nd)), dim3(TPB), 0, s, table_dev, nd, params, packed_bf, packed_f32, 0);
    return mmvae_check_launch("pack");
}
int launch_pack_range(const PackDesc* table_dev, const PackDesc* table_host, int nd, int d0, int d1, const float* params, bf16* packed_bf,
                      float* packed_f32, hipStream_t s) {
    MMVAE_REQUIRE(0 <= d0 && d0 <= d1 && d1 <= nd, "pack: descriptor range [%d, %d) of %d", d0, d1, nd);
    if (d0 == d1) return MMVAE_OK;
    const int b0 = (int)table_host[d0].first_block, b1 = d1 < nd ? (int)table_host[d1].first_block : table_blocks(table_host, nd);
    hipLaunchKernelGGL(pack_kernel, dim3(b1 - b0), dim3(TPB), 0, s, table_dev, nd, params, packed_bf, packed_f32, b0);
    return mmvae_check_launch("pack");
}
int launch_unpack_grads(const PackDesc* table_dev, const PackDesc* table_host, int nd, const float* gmat, const float* gvec,
                        float* grads, hipStream_t s, int part) {
    MMVAE_REQUIRE(nd > 0, "unpack: empty table");
    for (int i = 0; i < nd; ++i)
        MMVAE_REQUIRE((long long)table_host[i].Npad * (table_host[i].Kpad / 8) < (1 << 23), "unpack: matrix %d too large for 32-bit index math", i);
    hipLaunchKernelGGL(unpack_kernel<false>, dim3(table_blocks(table_host, nd)), dim3(TPB), 0, s, table_dev, nd, gmat, gvec, grads, (int*)nullptr, part);
    return mmvae_check_launch("unpack_grads");
}
int launch_unpack_map(const PackDesc* table_dev, const PackDesc* table_host, int nd, long long nparams, long long gmat_elems, int* map, hipStream_t s) {
    MMVAE_REQUIRE(nd > 0 && map, "unpack_map: empty table");
    MMVAE_REQUIRE(gmat_elems < (1ll << 31) - 2, "unpack_map: packed gradient buffer too large for 32-bit indices");
    if (hipMemsetAsync(map, 0xFF, (size_t)nparams * sizeof(int), s) != hipSuccess) { mmvae_set_error("hipMemsetAsync failed"); return MMVAE_EHIP; }
    hipLaunchKernelGGL(unpack_kernel<true>, dim3(table_blocks(table_host, nd)), dim3(TPB), 0, s, table_dev, nd, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, map, -1);
    return mmvae_check_launch("unpack_map");
}
int launch_im2col_small(const float* src, int Nimg, int Cin, int H, int W, int KH, int KW, int stride, int pad, int OH, int OW,
                        bf16* dst, int ld, hipStream_t s) {
    MMVAE_REQUIRE(ld % 8 == 0 && ld >= KH * KW * Cin && ld / 8 <= TPB, "im2col: ld=%d", ld);
    MMVAE_REQUIRE((long long)Nimg * OH * OW < (1 << 23) && (long long)Cin * H * W < (1ll << 31), "im2col: too many rows");
    const int rpb = TPB / (ld / 8);
    const long long nrows = (long long)Nimg * OH * OW;
    hipLaunchKernelGGL(im2col_small_kernel, dim3((unsigned)min((nrows + rpb - 1) / rpb, (long long)65536)), dim3(TPB), 0, s, src, Nimg, Cin, H, W,
                       KH, KW, stride, pad, OH, OW, dst, ld);
    return mmvae_check_launch("im2col_small");
}
int launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(!a.training || a.count > 1.f, "Expected more than 1 value per channel when training");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(a.C, 64)), dim3(64), 0, s, a);
    return mmvae_check_launch("bn_finalize");
}
int launch_bn_act(const BnActArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.ld % 8 == 0 && a.ld >= round_up(a.C, 8) && a.G * a.C <= 4096, "bn_act: C=%d ld=%d G=%d", a.C, a.ld, a.G);
    MMVAE_REQUIRE(!a.fin.training || a.fin.count > 1.f, "Expected more than 1 value per channel when training");
    hipLaunchKernelGGL(bn_act_kernel, dim3(nblocks((long long)a.rows * ((a.C + 7) / 8), TPB, 2048)), dim3(TPB),
                       (size_t)a.G * a.C * sizeof(float2), s, a);
    return mmvae_check_launch("bn_act");
}
int launch_bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.ld % 8 == 0 && a.ld >= round_up(a.C, 8), "bn_bwd_apply: C=%d ld=%d", a.C, a.ld);
    MMVAE_LAUNCH(bn_bwd_apply_kernel, dim3(nblocks((long long)a.rows * ((a.C + 7) / 8), TPB, 2048)), dim3(TPB),
                       (size_t)a.G * a.C * sizeof(float4), s, a);
    return mmvae_check_launch("bn_bwd_apply");
}
int launch_sigmoid_bce(const BceArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.G >= 1 && a.G <= 4, "bce: G=%d", a.G);
    long long per_group = (long long)a.B * a.C * a.H * a.W;
    hipLaunchKernelGGL(sigmoid_bce_kernel, dim3(nblocks(per_group, TPB * 4, 1024), a.G), dim3(TPB), 0, s, a);
    return mmvae_check_launch("sigmoid_bce");
}
int launch_poe_fwd(const float* mu, const float* lv, int M, int n, float* omu, float* olv, hipStream_t s) {
    hipLaunchKernelGGL(poe_fwd_kernel, dim3(ceil_div(n, TPB)), dim3(TPB), 0, s, mu, lv, M, n, omu, olv);
    return mmvae_check_launch("poe_fwd");
}
int launch_poe_bwd(const float* mu, const float* lv, int M, int n, const float* gmu, const float* glv, float* dmu, float* dlv, hipStream_t s) {
    hipLaunchKernelGGL(poe_bwd_kernel, dim3(ceil_div(n, TPB)), dim3(TPB), 0, s, mu, lv, M, n, gmu, glv, dmu, dlv);
    return mmvae_check_launch("poe_bwd");
}
int launch_reparam_fwd(const float* mu, const float* lv, const float* eps, int n, float* z, hipStream_t s) {
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ceil_div(n, TPB)), dim3(TPB), 0, s, mu, lv, eps, n, z);
    return mmvae_check_launch("reparam_fwd");
}
int launch_reparam_bwd(const float* lv, const float* eps, const float* dz, int n, float* dmu, float* dlv, hipStream_t s) {
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ceil_div(n, TPB)), dim3(TPB), 0, s, lv, eps, dz, n, dmu, dlv);
    return mmvae_check_launch("reparam_bwd");
}
int launch_kl_fwd(const float* mu, const float* lv, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(nblocks(n, TPB, 64)), dim3(TPB), 0, s, mu, lv, n, out);
    return mmvae_check_launch("kl_fwd");
}
int launch_kl_bwd(const float* mu, const float* lv, int n, float coef, const float* gs, float* dmu, float* dlv, hipStream_t s) {
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(ceil_div(n, TPB)), dim3(TPB), 0, s, mu, lv, n, coef, gs, dmu, dlv);
    return mmvae_check_launch("kl_bwd");
}
int launch_normal(float* out, long long n, unsigned long long seed, const long long* step, unsigned stream_id, hipStream_t s) {
    hipLaunchKernelGGL(normal_kernel, dim3(nblocks((n + 3) / 4, TPB, 512)), dim3(TPB), 0, s, out, n, seed, step, stream_id);
    return mmvae_check_launch("normal");
}
int launch_keep_mask(uint8_t* out, long long n, float p, unsigned long long seed, const long long* step, unsigned stream_id, hipStream_t s) {
    hipLaunchKernelGGL(keep_mask_kernel, dim3(nblocks((n + 3) / 4, TPB, 512)), dim3(TPB), 0, s, out, n, p, seed, step, stream_id);
    return mmvae_check_launch("keep_mask");
}
int launch_latent3_fwd(const Latent3Args& a, hipStream_t s) {
    MMVAE_REQUIRE(a.ldz >= a.D && a.ldz % 8 == 0, "latent3: ldz=%d", a.ldz);
    MMVAE_LAUNCH(latent3_fwd_kernel, dim3(nblocks((long long)a.B * a.ldz, TPB, 256)), dim3(TPB), 0, s, a);
    return mmvae_check_launch("latent3_fwd");
}
int launch_latent3_bwd(const Latent3BwdArgs& a, hipStream_t s) {
    MMVAE_REQUIRE((a.loss_slots == nullptr) == (a.loss_out == nullptr), "latent3_bwd: loss_slots / loss_out go together");
    MMVAE_LAUNCH(latent3_bwd_kernel, dim3(ceil_div(a.f.D, 64), ceil_div(a.f.B, L3B_ROWS)), dim3(TPB), 0, s, a);
    return mmvae_check_launch("latent3_bwd");
}
int launch_adam(const AdamArgs& a_in, hipStream_t s) {
    AdamArgs a = a_in;
    MMVAE_REQUIRE(a.step != nullptr, "adam: step counter is null");
    MMVAE_REQUIRE(a.nr >= 0 && a.nr <= 4, "adam: at most 4 element ranges");
    if (a.nr == 0) { a.nr = 1; a.roff[0] = 0; a.rlen[0] = a.n; }
    long long vecs = 0;
    for (int r = 0; r < a.nr; ++r) {
        MMVAE_REQUIRE(a.roff[r] >= 0 && a.rlen[r] >= 0 && a.roff[r] % 4 == 0 && a.roff[r] + a.rlen[r] <= a.n &&
                      (a.rlen[r] % 4 == 0 || a.roff[r] + a.rlen[r] == a.n), "adam: range %d = (%lld, %lld) of %lld elements", r, a.roff[r], a.rlen[r], a.n);
        vecs += (a.rlen[r] + 3) / 4;
    }
    // the word after the step counter is the block ticket (both live in the caller's 16-byte state block)
    unsigned* done = reinterpret_cast<unsigned*>(a.step + 1);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(vecs > 0 ? vecs : 1, TPB, mmvae_knob("adam_blocks", 512))), dim3(TPB), 0, s, a, done);   // few blocks: one ticket atomic each
    return mmvae_check_launch("adam");
}
static __global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, const long long* __restrict__ idx,
                                                                 long long row_bytes, int parts, char* __restrict__ dst) {
    const long long row = blockIdx.x / parts;
    const int part = blockIdx.x % parts;
    const long long vecs = row_bytes / 16, per = (vecs + parts - 1) / parts;
    const long long v0 = part * per, v1 = v0 + per < vecs ? v0 + per : vecs;
    const u32x4* s = reinterpret_cast<const u32x4*>(src + idx[row] * row_bytes);
    u32x4* d = reinterpret_cast<u32x4*>(dst + row * row_bytes);
    for (long long v = v0 + threadIdx.x; v < v1; v += 256) d[v] = s[v];
}
// rows of a multiple of 4 bytes (MultiMNIST: 2 500-byte images): one workgroup per row, 4-byte vectors
static __global__ __launch_bounds__(256) void gather_rows4_kernel(const char* __restrict__ src, const long long* __restrict__ idx,
                                                                  long long row_bytes, char* __restrict__ dst) {
    const long long row = blockIdx.x;
    const unsigned* s = reinterpret_cast<const unsigned*>(src + idx[row] * row_bytes);
    unsigned* d = reinterpret_cast<unsigned*>(dst + row * row_bytes);
    for (long long v = threadIdx.x; v < row_bytes / 4; v += 256) d[v] = s[v];
}
static __global__ __launch_bounds__(256) void gather_rows_u8_f32_kernel(const unsigned char* __restrict__ src, const long long* __restrict__ idx,
                                                                       long long row_elems, float denom, float* __restrict__ dst) {
    const long long row = blockIdx.x;
    const unsigned* s = reinterpret_cast<const unsigned*>(src + idx[row] * row_elems);
    f32x4* d = reinterpret_cast<f32x4*>(dst + row * row_elems);
    for (long long v = threadIdx.x; v < row_elems / 4; v += 256) {
        const unsigned w = s[v];
        d[v] = f32x4{(float)(w & 255u) / denom, (float)((w >> 8) & 255u) / denom, (float)((w >> 16) & 255u) / denom, (float)(w >> 24) / denom};     // IEEE division: bit-equal to ToTensor
    }
}
int launch_gather_rows_u8_f32(const unsigned char* src, const long long* idx, long long rows, long long row_elems, float denom, float* dst, hipStream_t s) {
    MMVAE_REQUIRE(src && idx && dst && rows >= 1 && rows < (1ll << 31) && row_elems >= 4 && row_elems % 4 == 0 && ((uintptr_t)src & 3) == 0 &&
                  ((uintptr_t)dst & 15) == 0, "gather_rows_u8_f32: rows of a multiple of 4 elements, aligned buffers");
    hipLaunchKernelGGL(gather_rows_u8_f32_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, idx, row_elems, denom, dst);
    return mmvae_check_launch("gather_rows_u8_f32");
}
int launch_gather_rows(const void* src, const long long* idx, long long rows, long long row_bytes, void* dst, hipStream_t s) {
    if (src && idx && dst && rows >= 1 && row_bytes >= 4 && row_bytes % 16 != 0) {
        MMVAE_REQUIRE(row_bytes % 4 == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0 && rows < (1ll << 31), "gather_rows: rows of a multiple of 4 bytes, 4-byte aligned buffers");
        hipLaunchKernelGGL(gather_rows4_kernel, dim3((unsigned)rows), dim3(256), 0, s, (const char*)src, idx, row_bytes, (char*)dst);
        return mmvae_check_launch("gather_rows");
    }
    MMVAE_REQUIRE(src && idx && dst && rows >= 1 && row_bytes >= 16 && row_bytes % 16 == 0, "gather_rows: rows of a multiple of 16 bytes");
    MMVAE_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "gather_rows: 16-byte aligned buffers");
    // enough workgroups in flight to cover the host link's latency: ~32 KB per workgroup
    int parts = (int)((row_bytes + 32767) / 32768);
    if (parts < 1) parts = 1;
    MMVAE_REQUIRE(rows * parts < (1ll << 31), "gather_rows: too many rows");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(rows * parts)), dim3(256), 0, s, (const char*)src, idx, row_bytes, parts, (char*)dst);
    return mmvae_check_launch("gather_rows");
}
static __global__ void step_losses_kernel(StepLossArgs a) {
    const int k = threadIdx.x;
    if (k < 3) a.out[k] = a.w_bce[k] * a.sums[k] + a.w_nll[k] * a.sums[4 + k] + a.w_kl[k] * a.sums[8 + k];
}
int launch_step_losses(const StepLossArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.sums && a.out, "step_losses: null buffer");
    hipLaunchKernelGGL(step_losses_kernel, dim3(1), dim3(64), 0, s, a);
    return mmvae_check_launch("step_losses");
}
int launch_fill_zero(void* p, size_t bytes, hipStream_t s) {
    if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) {
        mmvae_set_error("hipMemsetAsync failed");
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}

int launch_bce_fwd(const float* p, const float* t, long long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(nblocks(n, TPB * 4, 1024)), dim3(TPB), 0, s, p, t, n, out);
    return mmvae_check_launch("bce_fwd");
}
int launch_bce_bwd(const float* p, const float* t, long long n, float coef, const float* gs, float* dp, hipStream_t s) {
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, p, t, n, coef, gs, dp);
    return mmvae_check_launch("bce_bwd");
}
int launch_nll_fwd(const float* lp, const long long* tg, int rows, int classes, float* out, hipStream_t s) {
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(nblocks(rows, TPB, 64)), dim3(TPB), 0, s, lp, tg, rows, classes, out);
    return mmvae_check_launch("nll_fwd");
}
int launch_nll_bwd(const long long* tg, int rows, int classes, float coef, const float* gs, float* dlp, hipStream_t s) {
    hipLaunchKernelGGL(nll_bwd_kernel, dim3(ceil_div(rows * classes, TPB)), dim3(TPB), 0, s, tg, rows, classes, coef, gs, dlp);
    return mmvae_check_launch("nll_bwd");
}

int launch_step_begin(const StepBeginArgs& a, hipStream_t s) {
    for (int r = 0; r < 4; ++r)
        MMVAE_REQUIRE(a.zero_ptr[r] == nullptr || (a.zero_bytes[r] % 16 == 0 && ((uintptr_t)a.zero_ptr[r] & 15) == 0), "step_begin: zero range %d is not 16-byte aligned", r);
    MMVAE_LAUNCH(step_begin_kernel, dim3(mmvae_knob("begin_blocks", 2048) + (a.pack_blocks > 0 ? a.pack_blocks : 0)), dim3(TPB), 0, s, a);
    return mmvae_check_launch("step_begin");
}
// fills the pack part of a step prologue (StepBeginArgs::pack_*) from a descriptor table
int step_begin_with_pack(StepBeginArgs& a, const PackDesc* table_dev, const PackDesc* table_host, int nd, const float* params,
                         bf16* packed_bf, float* packed_f32) {
    MMVAE_REQUIRE(nd > 0, "pack: empty table");
    for (int i = 0; i < nd; ++i)
        MMVAE_REQUIRE((long long)table_host[i].Npad * (table_host[i].Kpad / 8) < (1 << 23), "pack: matrix %d too large for 32-bit index math", i);
    a.pack_table = table_dev; a.pack_nd = nd; a.pack_params = params; a.packed_bf = packed_bf; a.packed_f32 = packed_f32;
    a.pack_blocks = table_blocks(table_host, nd);
    return MMVAE_OK;
}

int launch_embed_gather_stats(const float* table, int C, const long long* idx, int rows, int idx_rows, int rows_per_group,
                              bf16* x, int ld, float2* colstats, hipStream_t s) {
    MMVAE_REQUIRE(ld >= C && ld % 8 == 0, "embed_gather: C=%d ld=%d", C, ld);
    hipLaunchKernelGGL(embed_gather_stats_kernel, dim3(ceil_div(rows, 64)), dim3(TPB), 0, s, table, C, idx, rows, idx_rows, rows_per_group, x, ld, colstats);
    return mmvae_check_launch("embed_gather_stats");
}
int launch_embed_scatter_add(const bf16* d, int ld, int C, const long long* idx, int rows, float* g_table, hipStream_t s) {
    hipLaunchKernelGGL(embed_scatter_add_kernel, dim3(ceil_div(rows * C, TPB)), dim3(TPB), 0, s, d, ld, C, idx, rows, g_table);
    return mmvae_check_launch("embed_scatter_add");
}
int launch_logsoftmax_nll(const LogSoftmaxNllArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(logsoftmax_nll_kernel, dim3(ceil_div(a.rows, TPB)), dim3(TPB), 0, s, a);
    return mmvae_check_launch("logsoftmax_nll");
}
int launch_cast_bf16(const float* x, long long n, bf16* out, hipStream_t s) {
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 4 * TPB - 1) / (4 * TPB))), dim3(TPB), 0, s, x, n, out);
    return mmvae_check_launch("cast_bf16");
}
int launch_colsum_f32(const float* x, int rows, int cols, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_kernel_t<float>, dim3(ceil_div(cols, 64), min(64, ceil_div(rows, 16))), dim3(TPB), 0, s, x, cols, rows, cols, out);
    return mmvae_check_launch("colsum_f32");
}
int launch_colsum_bf16(const bf16* x, int ld, int rows, int cols, float* out, hipStream_t s) {
    if (ld % 8 == 0 && ld / 8 <= TPB && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int RL = TPB / (ld / 8);
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3(max(1, min(256, ceil_div(rows, RL * 8)))), dim3(TPB), 0, s, x, ld, rows, cols, out);
        return mmvae_check_launch("colsum_bf16");
    }
    hipLaunchKernelGGL(colsum_kernel_t<bf16>, dim3(ceil_div(cols, 64), min(64, ceil_div(rows, 16))), dim3(TPB), 0, s, x, ld, rows, cols, out);
    return mmvae_check_launch("colsum_bf16");
}
