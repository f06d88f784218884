// gemm_small_kernel<MT, NT, KS>: implicit GEMMs whose pixel-row count is too small to fill 256 CUs with 128-row
// tiles (classifier / bottleneck / MLP layers and their data gradients: 0.1 .. 5 GFLOP, rows <= ~10k).
//
// The 128-row kernel gives such a problem a handful of workgroups and then needs split-K (two launches, fp32 slabs)
// or the 16-row weight-streaming kernel (a few dozen workgroups, each a serial chain over the whole weight matrix) to
// find parallelism; both run at a few TFLOP/s.  Here ONE WAVE owns one (16*MT) x (16*NT) output tile:
//   * both operands are loaded straight into MFMA fragment registers (the fragment layout of
//     v_mfma_f32_16x16x32_bf16 is a 16-byte load per lane), through buffer descriptors: 32-bit offsets and hardware
//     range checking (padded taps, rows past the end and weight rows past Npad read as zeros) -- no LDS, no barrier;
//   * a 64-deep chunk of fragments is in flight while the previous one feeds the MFMAs;
//   * the KS waves of a workgroup split the K range of the same tile and are summed through LDS once (in-kernel
//     split-K: one launch, no global slabs);
//   * the product is computed transposed (weights in the A slot), so a lane owns 4 consecutive channels of one of its
//     own gathered pixel rows: the full epilogue (bias, bf16/fp32 stores, activated copy with dropout mask,
//     BatchNorm statistics, d-activation with BatchNorm-backward sums, bias-gradient column sums) runs from registers
//     with 8/16-byte accesses.
#include "gemm.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ bf16x8 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
    return __builtin_bit_cast(bf16x8, v);
}

// q = r / d, rem = r % d for 0 <= r < 2^23 with inv = 1.0f/d (one correction step each way)
__device__ __forceinline__ int fast_divmod(int r, int d, float inv, int& rem) {
    int q = (int)((float)r * inv);
    rem = r - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}

template <int MT, int NT, int KS, bool DGRAD>
__global__ __launch_bounds__(KS * 64) void gemm_small_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GatherCommon& c = p.c;
    const GatherClass& k = p.cls[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int RW = MT * 16, CW = NT * 16;
    const int tiles_per_group = (k.rows_per_group + RW - 1) / RW;
    if ((int)blockIdx.x >= tiles_per_group * c.groups) return;
    const int g = blockIdx.x / tiles_per_group;
    const int row0 = (blockIdx.x - g * tiles_per_group) * RW;
    const int n_base = blockIdx.y * CW;
    const int K = k.K, Kpad = k.Kpad;
    const int nch_all = (K + 63) / 64;
    const int ch0 = nch_all * wave / KS, ch1 = nch_all * (wave + 1) / KS;

    const int nimg_a = c.a_bcast_n > 0 ? c.a_bcast_n : c.groups * c.group_n;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(c.A), 0, (int)((size_t)nimg_a * c.AH * c.AW * c.Ald * sizeof(bf16)), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(k.Wp), 0, (int)((size_t)p.npad * Kpad * sizeof(bf16)), 0x00020000);

    // ---- this lane's pixel rows (operand fragment layout: row = mt*16 + lane%16)
    const int pix_per_img = k.OY * k.OX;
    const float inv_pix = 1.0f / (float)pix_per_img, inv_ox = 1.0f / (float)k.OX, inv_tw = 1.0f / (float)k.TW;
    int a_off[MT], opix[MT], dpix[MT], grow[MT];
    unsigned mtap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = row0 + mt * 16 + fr;
        a_off[mt] = 0; mtap[mt] = 0; opix[mt] = -1; dpix[mt] = 0; grow[mt] = 0;
        if (r < k.rows_per_group) {
            int rem, ox;
            const int img = fast_divmod(r, pix_per_img, inv_pix, rem);
            const int oy = fast_divmod(rem, k.OX, inv_ox, ox);
            const int nimg = g * c.group_n + img;
            int aimg = nimg;
            if (c.a_bcast_n > 0) aimg %= c.a_bcast_n;
            const int y0 = oy * c.sy + k.offy, x0 = ox * c.sx + k.offx;
            a_off[mt] = (((aimg * c.AH + y0) * c.AW + x0) * c.Ald + fq * 8) * (int)sizeof(bf16);
            unsigned mx = 0, m = 0;
            for (int tx = 0; tx < k.TW; ++tx) mx |= ((unsigned)(x0 + tx * c.dx) < (unsigned)c.AW ? 1u : 0u) << tx;
            for (int ty = 0; ty < k.TH; ++ty)
                if ((unsigned)(y0 + ty * c.dy) < (unsigned)c.AH) m |= mx << (ty * k.TW);
            mtap[mt] = m;
            const int py = oy * c.osy + k.ooy, px = ox * c.osx + k.oox;
            opix[mt] = (nimg * c.OH + py) * c.OW + px;
            dpix[mt] = p.d_bcast_n > 0 ? ((nimg % p.d_bcast_n) * c.OH + py) * c.OW + px : opix[mt];
            grow[mt] = g * k.rows_per_group + r;
        }
    }
    const int ntaps = k.TH * k.TW;
    const int cshift = 31 - __clz(c.C);          // multi-tap operands have a power-of-two channel count >= 32 (launcher)
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_off[nt] = ((n_base + nt * 16 + fr) * Kpad + fq * 8) * (int)sizeof(bf16);

    bf16x8 a0[2][MT], a1[2][MT], b0[2][NT], b1[2][NT];
    auto load_chunk = [&](int ch, bf16x8 (&a)[2][MT], bf16x8 (&b)[2][NT]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kk0 = ch * 64 + ks * 32;
            int tap = 0, acb = kk0;
            if (ntaps > 1) { tap = kk0 >> cshift; acb = kk0 & (c.C - 1); }    // wave-uniform
            const int ty = (int)(((float)tap + 0.5f) * inv_tw), tx = tap - ty * k.TW;
            const int toff = (((ty * c.dy) * c.AW + tx * c.dx) * c.Ald + acb) * (int)sizeof(bf16);
            const bool kin = kk0 + fq * 8 < K;                               // only the last k-step can be partial
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned bit = kin ? (mtap[mt] >> tap) & 1u : 0u;
                a[ks][mt] = buf_load16(arsrc, (unsigned)(a_off[mt] + toff) | (bit - 1u));   // invalid -> 0xFFFFFFFF -> zeros
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[ks][nt] = buf_load16(brsrc, kk0 < Kpad ? (unsigned)(b_off[nt] + kk0 * (int)sizeof(bf16)) : 0xFFFFFFFFu);
        }
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute_chunk = [&](bf16x8 (&a)[2][MT], bf16x8 (&b)[2][NT]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)           // transposed product: rows = channels, columns = pixel rows
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][nt], a[ks][mt], acc[mt][nt], 0, 0, 0);
    };

    if (ch0 < ch1) load_chunk(ch0, a0, b0);
    for (int ch = ch0; ch < ch1; ch += 2) {
        if (ch + 1 < ch1) load_chunk(ch + 1, a1, b1);
        compute_chunk(a0, b0);
        if (ch + 1 < ch1) {
            if (ch + 2 < ch1) load_chunk(ch + 2, a0, b0);
            compute_chunk(a1, b1);
        }
    }

    // ---- in-kernel split-K: waves 1.. hand their partial tile to wave 0 through LDS
    if (KS > 1) {
        float* red = reinterpret_cast<float*>(smem);          // [KS-1][MT*NT*4][64]
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        red[((wave - 1) * (MT * NT * 4) + (mt * NT + nt) * 4 + j) * 64 + lane] = acc[mt][nt][j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < KS - 1; ++w)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mt][nt][j] += red[(w * (MT * NT * 4) + (mt * NT + nt) * 4 + j) * 64 + lane];
    }

    // ---- epilogue: lane owns channels n_base + nt*16 + fq*4 .. +3 of pixel row mt*16 + fr
    const bool want_stats = !DGRAD && p.colstats != nullptr;
    const bool want_red = DGRAD && (p.d_red != nullptr || p.d_colsum != nullptr);
    const int tn = p.d_cmod > 0 ? p.d_cmod : c.N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = n_base + nt * 16 + fq * 4;
        if (n0 >= c.N) continue;
        const int nv = min(4, c.N - n0);                      // valid channels of this quad
        float bias4[4], dsc[4], dsh[4], dmean[4], drstd[4], s1[4], s2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = min(n0 + j, c.N - 1);
            const int tcol = p.d_cmod > 0 ? n % p.d_cmod : n;
            bias4[j] = p.bias ? p.bias[n] : 0.f;
            dsc[j] = 1.f; dsh[j] = 0.f; dmean[j] = 0.f; drstd[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f;
            if (DGRAD && p.d_affine) { const float2 a = p.d_affine[g * tn + tcol]; dsc[j] = a.x; dsh[j] = a.y; }
            if (DGRAD && p.d_meanrstd) { const float2 m = p.d_meanrstd[g * tn + tcol]; dmean[j] = m.x; drstd[j] = m.y; }
        }
        const bool vec = nv == 4 && p.ldo % 4 == 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (opix[mt] < 0) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] + bias4[j];
            if (DGRAD) {
                float rr[4];
                const bf16* rp = p.d_r + (size_t)dpix[mt] * p.d_ld + n0;
                if (nv == 4 && p.d_ld % 4 == 0) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rp);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rr[j] = (float)rv[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rr[j] = j < nv ? (float)rp[j] : 0.f;
                }
                unsigned mb = 0x01010101u;
                if (p.d_mask) {
                    const uint8_t* mp = p.d_mask + (size_t)grow[mt] * c.N + n0;
                    if (nv == 4 && c.N % 4 == 0) mb = *reinterpret_cast<const unsigned*>(mp);
                    else { mb = 0; for (int j = 0; j < nv; ++j) mb |= (unsigned)(mp[j] ? 1 : 0) << (8 * j); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j] * act_bwd(p.d_act, rr[j] * dsc[j] + dsh[j]);
                    if (p.d_mask) x = ((mb >> (8 * j)) & 0xff) ? x * p.d_mask_scale : 0.f;
                    v[j] = x;
                    if (want_red && j < nv) { s1[j] += x; s2[j] += x * (rr[j] - dmean[j]) * drstd[j]; }
                }
            }
            if (want_stats) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
            }
            float av[4];
            if (!DGRAD && p.out_act_bf) {
                unsigned mb = 0x01010101u;
                if (p.e_mask) {
                    const uint8_t* mp = p.e_mask + (size_t)grow[mt] * c.N + n0;
                    if (nv == 4 && c.N % 4 == 0) mb = *reinterpret_cast<const unsigned*>(mp);
                    else { mb = 0; for (int j = 0; j < nv; ++j) mb |= (unsigned)(mp[j] ? 1 : 0) << (8 * j); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = act_fwd(p.e_act, v[j]);
                    if (p.e_mask) x = ((mb >> (8 * j)) & 0xff) ? x * p.e_mask_scale : 0.f;
                    av[j] = x;
                }
            }
            const size_t o = (size_t)opix[mt] * p.ldo + n0;
            if (vec) {
                if (p.out_bf) {
                    bf16x4 ob;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ob[j] = (bf16)v[j];
                    *reinterpret_cast<bf16x4*>(p.out_bf + o) = ob;
                }
                if (!DGRAD && p.out_act_bf) {
                    bf16x4 ob;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ob[j] = (bf16)av[j];
                    *reinterpret_cast<bf16x4*>(p.out_act_bf + o) = ob;
                }
                if (p.out_f) *reinterpret_cast<f32x4*>(p.out_f + o) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                for (int j = 0; j < nv; ++j) {
                    if (p.out_bf) p.out_bf[o + j] = (bf16)v[j];
                    if (!DGRAD && p.out_act_bf) p.out_act_bf[o + j] = (bf16)av[j];
                    if (p.out_f) p.out_f[o + j] = v[j];
                }
            }
        }
        if (want_stats || want_red) {
            const int slot = (blockIdx.x + 5 * blockIdx.z) % MMVAE_STAT_SLOTS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = s1[j], b = s2[j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (fr == 0 && j < nv) {
                    const int n = n0 + j;
                    float2* dst = want_stats ? p.colstats : p.d_red;
                    if (dst) {
                        const bool cm = !want_stats && p.d_cmod > 0;
                        const int tcol = cm ? n % p.d_cmod : n;
                        const int sl = (slot + (cm ? n / p.d_cmod : 0)) % MMVAE_STAT_SLOTS;
                        atomicAdd(&dst[(g * MMVAE_STAT_SLOTS + sl) * (cm ? p.d_cmod : c.N) + tcol].x, a);
                        atomicAdd(&dst[(g * MMVAE_STAT_SLOTS + sl) * (cm ? p.d_cmod : c.N) + tcol].y, b);
                    }
                    if (p.d_colsum) atomicAdd(p.d_colsum + n, a);
                }
            }
        }
    }
}

template <int MT, int NT, int KS, bool DGRAD>
int launch_small(const GemmParams& p, hipStream_t stream) {
    const GatherCommon& c = p.c;
    int max_tiles = 0;
    for (int i = 0; i < c.nclasses; ++i) max_tiles = max(max_tiles, ceil_div(p.cls[i].rows_per_group, MT * 16));
    dim3 grid(max_tiles * c.groups, ceil_div(c.N, NT * 16), c.nclasses);
    const size_t lds = KS > 1 ? (size_t)(KS - 1) * MT * NT * 4 * 64 * sizeof(float) : 0;
    MMVAE_LAUNCH((gemm_small_kernel<MT, NT, KS, DGRAD>), grid, dim3(KS * 64), lds, stream, p);
    return mmvae_check_launch("gemm_small");
}

template <int MT, int NT, bool DGRAD>
int launch_small_ks(const GemmParams& p, int ks, hipStream_t stream) {
    if (ks >= 4) return launch_small<MT, NT, 4, DGRAD>(p, stream);
    if (ks >= 2) return launch_small<MT, NT, 2, DGRAD>(p, stream);
    return launch_small<MT, NT, 1, DGRAD>(p, stream);
}

}  // namespace

// returns 1 when the problem was launched here, 0 when the caller should use the generic kernels, < 0 on error
int try_launch_gemm_small(const GemmParams& pin, hipStream_t stream) {
    const GatherCommon& c = pin.c;
    int max_tiles128 = 0, min_nch = 1 << 30;
    long long tiles22 = 0, tiles44 = 0, rows_total = 0;
    for (int i = 0; i < c.nclasses; ++i) {
        const GatherClass& k = pin.cls[i];
        if (k.TH * k.TW > 1 && (c.C < 32 || (c.C & (c.C - 1)) != 0)) return 0;   // the tap walk must be wave-uniform
        if (k.TH * k.TW > 32) return 0;                                           // tap-validity bit mask
        max_tiles128 = max(max_tiles128, ceil_div(k.rows_per_group, 128));
        min_nch = min(min_nch, ceil_div(k.K, 64));
        tiles22 += (long long)ceil_div(k.rows_per_group, 32) * c.groups * ceil_div(c.N, 32);
        tiles44 += (long long)ceil_div(k.rows_per_group, 64) * c.groups * ceil_div(c.N, 64);
        rows_total += (long long)k.rows_per_group * c.groups;
    }
    // only problems that leave most of the chip idle under 128-row tiles
    const long long tiles128 = (long long)max_tiles128 * c.groups * c.nclasses * ceil_div(c.N, 128);
    if (tiles128 > 128) return 0;
    {   // ... and not the long-K giants: one wave per 32x32 tile re-reads both operands from L2 with no reuse
        constexpr double cap = 8.0;   // CelebA classifier.0 (13.4 GFLOP): 107 us here, ~35 us tiled + split-K
        double fl = 0.0;
        for (int i = 0; i < c.nclasses; ++i) fl += 2.0 * pin.cls[i].rows_per_group * c.groups * c.N * pin.cls[i].K;
        if (cap > 0.0 && fl > cap * 1e9) return 0;
    }
    const long long nimg_a = c.a_bcast_n > 0 ? c.a_bcast_n : (long long)c.groups * c.group_n;
    if (nimg_a * c.AH * c.AW * c.Ald * 2 >= (1ll << 31)) return 0;                 // 32-bit buffer offsets
    if ((long long)c.groups * c.group_n * c.OH * c.OW >= (1ll << 31) || rows_total >= (1ll << 23)) return 0;
    GemmParams p = pin;
    p.ksplit = 1; p.sk_buf = nullptr;
    // packed rows available (layers.h npad_for); the kernel range-checks weight rows against it
    if (p.npad <= 0) p.npad = c.N <= 16 ? 16 : c.N <= 32 ? 32 : c.N <= 64 ? 64 : round_up(c.N, 128);
    (void)tiles44;
    if (tiles22 > 1536) return 0;                                                  // enough work for the tile kernels
    if (p.d_r == nullptr && (p.d_mask || p.d_red || p.d_colsum)) return 0;
    constexpr long long wave_target = 1024;
    const int ks = (int)min((long long)min(4, max(1, min_nch / 2)), max(1ll, wave_target / max(tiles22, 1ll)));
    const int rc = p.d_r ? launch_small_ks<2, 2, true>(p, ks, stream) : launch_small_ks<2, 2, false>(p, ks, stream);
    return rc == MMVAE_OK ? 1 : rc;
}
