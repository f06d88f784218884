// Device pieces shared by the MFMA implicit-GEMM kernels (gemm.hip: gathered k-tiles staged through LDS; gemm_img.hip:
// whole images staged in LDS): reciprocal division, operand transform and the common epilogue.
#pragma once
#include "gemm.h"

namespace {

// q = r / d, rem = r % d for 0 <= r < 2^23 with inv = 1.0f/d (one correction step each way): ~7 VALU instructions
// instead of the ~35 of an integer division
__device__ __forceinline__ int fast_divmod(int r, int d, float inv, int& rem) {
    int q = (int)((float)r * inv);
    rem = r - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}

struct RowCoord {
    int pix;    // n*AH*AW (gather base) -- or -1 when the row is out of range
    int y, x;   // oy*sy+offy, ox*sx+offx
};

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
    return z;
}

// applies (affine, act, keep-mask) to 8 gathered channels
__device__ __forceinline__ bf16x8 transform8(bf16x8 v, const float2* aff, int act, const uint8_t* mask, float mscale) {
    bf16x8 o;
    uint64_t mbits = 0;
    if (mask) mbits = *reinterpret_cast<const uint64_t*>(mask);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float f = (float)v[j];
        if (aff) f = f * aff[j].x + aff[j].y;
        f = act_fwd(act, f);
        if (mask) f = ((mbits >> (8 * j)) & 0xff) ? f * mscale : 0.f;
        o[j] = (bf16)f;
    }
    return o;
}

// Shared epilogue: consumes the fp32 tile `ct` ([BM][BN+4] in LDS) of output rows row0.. / columns n0..
// `row_end`: rows >= row_end are not this workgroup's (gemm_img: the tile ends with the workgroup's last image)
template <int NT, int ROWS>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const GatherClass& k, int g, int row0, int n0,
                                              float* ct, char* smem, int tid, int row_end = 0x7fffffff) {
    constexpr int BN = NT * 16;
    constexpr int LDC = BN + 4;
    constexpr int VPR = BN / 8;              // vectors per tile row
    constexpr int RPP = 256 / VPR;           // tile rows per pass
    constexpr int PASSES = ROWS / RPP;
    const GatherCommon& c = p.c;
    const int pix_per_img = k.OY * k.OX;
    const float inv_pix = 1.0f / (float)pix_per_img, inv_ox = 1.0f / (float)k.OX;
    const bool want_stats = p.colstats != nullptr;
    const bool want_red = p.d_red != nullptr || p.d_colsum != nullptr;
    const int cv = tid % VPR;
    const int col0 = n0 + cv * 8;
    const bool vec_ok = (col0 + 8 <= c.N) && (p.ldo % 8 == 0);
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    float bias8[8], dsc[8], dsh[8], dmean[8], drstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = min(col0 + j, c.N - 1);
        bias8[j] = p.bias ? p.bias[col] : 0.f;
        dsc[j] = 1.f; dsh[j] = 0.f; dmean[j] = 0.f; drstd[j] = 0.f;
        const int tn = p.d_cmod > 0 ? p.d_cmod : c.N, tcol = p.d_cmod > 0 ? col % p.d_cmod : col;
        if (p.d_affine) { float2 a = p.d_affine[g * tn + tcol]; dsc[j] = a.x; dsh[j] = a.y; }
        if (p.d_meanrstd) { float2 m = p.d_meanrstd[g * tn + tcol]; dmean[j] = m.x; drstd[j] = m.y; }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int rl = ps * RPP + tid / VPR;
        const int r = row0 + rl;
        if (r >= k.rows_per_group || r >= row_end || col0 >= c.N) continue;
        int rem, ox;
        const int img = fast_divmod(r, pix_per_img, inv_pix, rem);
        const int oy = fast_divmod(rem, k.OX, inv_ox, ox);
        const int nimg = g * c.group_n + img;
        const size_t opix = (size_t)(nimg * c.OH + oy * c.osy + k.ooy) * c.OW + ox * c.osx + k.oox;
        const size_t growi = (size_t)g * k.rows_per_group + r;
        float v[8];
        {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + rl * LDC + cv * 8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + rl * LDC + cv * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j] + bias8[j]; v[4 + j] = hi[j] + bias8[4 + j]; }
        }
        if (p.d_r) {
            size_t rpix = opix;
            if (p.d_bcast_n > 0) rpix = (size_t)((nimg % p.d_bcast_n) * c.OH + oy * c.osy + k.ooy) * c.OW + ox * c.osx + k.oox;
            float rr[8];
            if (vec_ok && p.d_ld % 8 == 0) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(p.d_r + rpix * p.d_ld + col0);
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = (float)rv[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = (col0 + j < c.N) ? (float)p.d_r[rpix * p.d_ld + col0 + j] : 0.f;
            }
            uint64_t mb = ~0ull;
            if (p.d_mask) {
                if (vec_ok) mb = *reinterpret_cast<const uint64_t*>(p.d_mask + growi * c.N + col0);
                else {
                    mb = 0;
                    for (int j = 0; j < 8; ++j)
                        if (col0 + j < c.N && p.d_mask[growi * c.N + col0 + j]) mb |= 0xffull << (8 * j);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = v[j] * act_bwd(p.d_act, rr[j] * dsc[j] + dsh[j]);
                if (p.d_mask) x = ((mb >> (8 * j)) & 0xff) ? x * p.d_mask_scale : 0.f;
                v[j] = x;
                if (want_red && col0 + j < c.N) { s1[j] += x; s2[j] += x * (rr[j] - dmean[j]) * drstd[j]; }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (col0 + j < c.N) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
        float av[8];
        if (p.out_act_bf) {
            uint64_t mb = ~0ull;
            if (p.e_mask) {
                if (vec_ok) mb = *reinterpret_cast<const uint64_t*>(p.e_mask + growi * c.N + col0);
                else {
                    mb = 0;
                    for (int j = 0; j < 8; ++j)
                        if (col0 + j < c.N && p.e_mask[growi * c.N + col0 + j]) mb |= 0xffull << (8 * j);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = act_fwd(p.e_act, v[j]);
                if (p.e_mask) x = ((mb >> (8 * j)) & 0xff) ? x * p.e_mask_scale : 0.f;
                av[j] = x;
            }
        }
        if (vec_ok) {
            if (p.out_bf) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
                *reinterpret_cast<bf16x8*>(p.out_bf + opix * p.ldo + col0) = o;
            }
            if (p.out_act_bf) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)av[j];
                *reinterpret_cast<bf16x8*>(p.out_act_bf + opix * p.ldo + col0) = o;
            }
            if (p.out_f) {
                *reinterpret_cast<f32x4*>(p.out_f + opix * p.ldo + col0) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(p.out_f + opix * p.ldo + col0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        } else {
            for (int j = 0; j < 8; ++j) {
                if (col0 + j >= c.N) break;
                if (p.out_bf) p.out_bf[opix * p.ldo + col0 + j] = (bf16)v[j];
                if (p.out_act_bf) p.out_act_bf[opix * p.ldo + col0 + j] = (bf16)av[j];
                if (p.out_f) p.out_f[opix * p.ldo + col0 + j] = v[j];
            }
        }
    }
    if (want_stats || want_red) {
        __syncthreads();                                       // everyone is done reading ct
        float2* red = reinterpret_cast<float2*>(smem);         // [RPP][BN]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid / VPR) * BN + cv * 8 + j] = make_float2(s1[j], s2[j]);
        __syncthreads();
        // two-level column sum: all 256 threads fold RPP rows down to PARTS partial rows, then BN threads finish
        // (a single pass left BN threads walking RPP = 64 rows for the narrow tiles while 7/8 of the workgroup idled)
        constexpr int PARTS = 256 / BN;
        {
            const int col = tid % BN, part = tid / BN;
            float a = 0.f, b = 0.f;
            for (int q = part; q < RPP; q += PARTS) { a += red[q * BN + col].x; b += red[q * BN + col].y; }
            __syncthreads();
            red[part * BN + col] = make_float2(a, b);
        }
        __syncthreads();
        if (tid < BN && n0 + tid < c.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) { a += red[q * BN + tid].x; b += red[q * BN + tid].y; }
            float2* dst = want_stats ? p.colstats : p.d_red;      // [groups][SLOTS][N]
            if (dst) {
                const bool cm = !want_stats && p.d_cmod > 0;
                const int tn = cm ? p.d_cmod : c.N, tcol = cm ? (n0 + tid) % p.d_cmod : n0 + tid;
                const int slot = (blockIdx.x + 5 * blockIdx.z + (cm ? (n0 + tid) / p.d_cmod : 0)) % MMVAE_STAT_SLOTS;
                atomicAdd(&dst[(g * MMVAE_STAT_SLOTS + slot) * tn + tcol].x, a);
                atomicAdd(&dst[(g * MMVAE_STAT_SLOTS + slot) * tn + tcol].y, b);
            }
            if (p.d_colsum) atomicAdd(p.d_colsum + n0 + tid, a);
        }
    }
}

}  // namespace
