// gemm_direct_kernel<NT, MT, NW, DGRAD>: the narrow-output (N <= 64) implicit GEMMs of the image layers -- the
// widest pixel grids of the models (conv1/conv2, the last transposed convs and their data gradients).
//
// Why a second kernel.  With 16..64 output columns every gathered A element feeds only N MACs; the 128-row tile
// kernel (gemm.hip) then spends its time on everything BUT the MFMAs: per-vector 64-bit address arithmetic and
// bounds selects, the A tile's round trip through LDS, one barrier and one exposed load latency per 64-deep k-tile,
// and a workgroup-wide LDS epilogue with integer divisions per output vector.  Here:
//   * A never touches LDS: the operand-fragment layout of v_mfma_f32_16x16x32_bf16 (lane -> index lane%16, 8
//     consecutive k at 8*(lane/16)) IS a 16-byte load per lane, so each wave gathers its own pixel rows straight into
//     fragment registers, one 64-deep chunk ahead of the MFMAs (and the next tile's first chunk ahead of the epilogue).
//   * the gather goes through a buffer descriptor: 32-bit offsets, and a padded / out-of-range tap is an offset of
//     0xFFFFFFFF that the hardware range check turns into zeros -- no clamped address, no select, no validity state.
//     Tap validity per pixel row is a bit mask computed once per tile; the tap walk itself is wave-uniform (scalar).
//   * B (the whole packed weight matrix of the class, [N][Kpad] bf16 <= ~100 KB) is staged in LDS ONCE per
//     workgroup; the k-loop has no barrier.  Rows are padded by 16 B -> conflict-free ds_read_b128.
//   * the product is computed TRANSPOSED (weights in the A slot, pixels in the B slot): a lane then owns 4 consecutive
//     channels of one of ITS OWN gathered pixel rows, so the epilogue needs no LDS, no barrier and no new coordinates:
//     8-byte bf16x4 stores, BatchNorm statistics accumulated in registers across all tiles and flushed once.
//   * workgroups are persistent over row tiles of their class (weights stay resident) and classes get workgroups in
//     proportion to their work.
#include "gemm.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16)0.0f;
    return z;
}

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

struct RowSet {          // per pixel row of this lane (row = mt*16 + lane%16)
    int a_off;           // byte offset of tap (0,0), channel 8*(lane/16) in the gathered tensor (may be "negative")
    unsigned mtap;       // bit ty*TW+tx set when that tap lies inside the gathered tensor (0 for rows past the end)
    int opix;            // pixel index in the output tensor, -1 for rows past the end
};

template <int NT, int MT, int NW, bool DGRAD>
__global__ __launch_bounds__(NW * 64) void gemm_direct_kernel(const GemmParams p) {
    constexpr int BN = NT * 16;
    constexpr int RW = MT * 16;              // rows per wave
    constexpr int BMD = NW * RW;             // rows per workgroup tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const GatherCommon& c = p.c;
    if (p.dbg & 16) return;
    // class of this workgroup (p.dblk = prefix sums of workgroups per class)
    int ci = 0;
#pragma unroll
    for (int i = 1; i < MMVAE_MAX_CLASSES; ++i)
        if (i < c.nclasses && (int)blockIdx.x >= p.dblk[i]) ci = i;
    const GatherClass& k = p.cls[ci];
    const int blk = blockIdx.x - p.dblk[ci], nblk = p.dblk[ci + 1] - p.dblk[ci];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = k.K, Kpad = k.Kpad;
    const int LDB = Kpad + 8;
    bf16* Bs = reinterpret_cast<bf16*>(smem);                                    // [BN][LDB]
    int* tapoff = reinterpret_cast<int*>(smem + (size_t)BN * LDB * sizeof(bf16));        // [32] byte offset of tap (ty,tx)
    float4* tabs = reinterpret_cast<float4*>(tapoff + 32);                               // DGRAD: [groups][BN] (sc, sh, mean, rstd)

    // ---- stage the packed weights of the class once (+ the BatchNorm tables of the d-activation)
    {
        const int vpr = Kpad / 8;
        for (int v = tid; v < BN * vpr; v += NW * 64) {
            const int n = v / vpr, kv = v - n * vpr;
            *reinterpret_cast<bf16x8*>(Bs + n * LDB + kv * 8) = *reinterpret_cast<const bf16x8*>(k.Wp + (size_t)n * Kpad + kv * 8);
        }
        if (tid < k.TH * k.TW) {
            const int ty = tid / k.TW, tx = tid - ty * k.TW;
            tapoff[tid] = (((ty * c.dy) * c.AW + tx * c.dx) * c.Ald) * (int)sizeof(bf16);
        }
        if (DGRAD) {
            for (int v = tid; v < c.groups * BN; v += NW * 64) {
                const int g = v / BN, n = v - g * BN;
                float4 t = make_float4(1.f, 0.f, 0.f, 0.f);
                if (n < c.N) {
                    if (p.d_affine) { const float2 a = p.d_affine[g * c.N + n]; t.x = a.x; t.y = a.y; }
                    if (p.d_meanrstd) { const float2 m = p.d_meanrstd[g * c.N + n]; t.z = m.x; t.w = m.y; }
                }
                tabs[v] = t;
            }
        }
    }
    __syncthreads();
    if (p.dbg & 8) return;

    const int nimg_a = c.a_bcast_n > 0 ? c.a_bcast_n : c.groups * c.group_n;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(c.A), 0, (int)((size_t)nimg_a * c.AH * c.AW * c.Ald * sizeof(bf16)), 0x00020000);

    const int pix_per_img = k.OY * k.OX;
    const float inv_pix = 1.0f / (float)pix_per_img, inv_ox = 1.0f / (float)k.OX;
    const int tiles_per_group = (k.rows_per_group + BMD - 1) / BMD;
    const int ntiles = tiles_per_group * c.groups;
    const int nch = (K + 63) / 64;
    const int ntaps = k.TH * k.TW;
    const int cshift = 31 - __clz(c.C);          // multi-tap operands have a power-of-two channel count (launcher)
    const bool want_stats = p.colstats != nullptr;
    const bool want_red = DGRAD && p.d_red != nullptr;
    float s1[NT][4], s2[NT][4];              // statistics of channels nt*16 + fq*4 + j over this lane's pixel rows
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nt][j] = 0.f; s2[nt][j] = 0.f; }
    int g_acc = -1;                          // group the register statistics belong to

    auto flush_stats = [&]() {
        if (g_acc < 0 || !(want_stats || want_red)) return;
        float2* dst = want_stats ? p.colstats : p.d_red;
        const int slot = (blockIdx.x * NW + wave) % MMVAE_STAT_SLOTS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = s1[nt][j], b = s2[nt][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                const int n = nt * 16 + fq * 4 + j;
                if (fr == 0 && n < c.N) {
                    atomicAdd(&dst[(g_acc * MMVAE_STAT_SLOTS + slot) * c.N + n].x, a);
                    atomicAdd(&dst[(g_acc * MMVAE_STAT_SLOTS + slot) * c.N + n].y, b);
                }
                s1[nt][j] = 0.f; s2[nt][j] = 0.f;
            }
    };

    // coordinates of this lane's pixel rows in tile t
    auto tile_rows = [&](int t, RowSet (&rs)[MT]) {
        const bool live = t < ntiles;
        const int g = live ? t / tiles_per_group : 0;
        const int row0 = (t - g * tiles_per_group) * BMD + wave * RW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = row0 + mt * 16 + fr;
            rs[mt].a_off = 0; rs[mt].mtap = 0; rs[mt].opix = -1;
            if (live && r < k.rows_per_group) {
                int rem, ox;
                const int img = fast_divmod(r, pix_per_img, inv_pix, rem);
                const int oy = fast_divmod(rem, k.OX, inv_ox, ox);
                const int nimg = g * c.group_n + img;
                int aimg = nimg;
                if (c.a_bcast_n > 0) aimg %= c.a_bcast_n;
                const int y0 = oy * c.sy + k.offy, x0 = ox * c.sx + k.offx;
                rs[mt].a_off = (((aimg * c.AH + y0) * c.AW + x0) * c.Ald + fq * 8) * (int)sizeof(bf16);
                unsigned mx = 0, m = 0;
                for (int tx = 0; tx < k.TW; ++tx) mx |= ((unsigned)(x0 + tx * c.dx) < (unsigned)c.AW ? 1u : 0u) << tx;
                for (int ty = 0; ty < k.TH; ++ty)
                    if ((unsigned)(y0 + ty * c.dy) < (unsigned)c.AH) m |= mx << (ty * k.TW);
                rs[mt].mtap = m;
                rs[mt].opix = (nimg * c.OH + oy * c.osy + k.ooy) * c.OW + ox * c.osx + k.oox;
            }
        }
    };

    // one 64-deep chunk of A fragments (2 MFMA k-steps x MT pixel-row tiles); the tap walk is wave-uniform
    auto load_chunk = [&](int ch, const RowSet (&rs)[MT], bf16x8 (&a)[2][MT]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kk0 = ch * 64 + ks * 32;
            int tap = 0, acb = kk0;
            if (ntaps > 1) { tap = kk0 >> cshift; acb = kk0 & (c.C - 1); }    // uniform: C is a power of two >= 32 here
            const int toff = tapoff[min(tap, 31)] + acb * (int)sizeof(bf16);
            const bool kin = kk0 + fq * 8 < K;                               // only the last k-step can be partial
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned bit = kin ? (rs[mt].mtap >> tap) & 1u : 0u;
                const unsigned off = (unsigned)(rs[mt].a_off + toff) | (bit - 1u);   // invalid -> 0xFFFFFFFF -> zeros
                if (p.dbg & 1) { a[ks][mt] = zero8(); a[ks][mt][0] = (bf16)(float)off; }
                else a[ks][mt] = buf_load16(arsrc, off);
            }
        }
    };

    RowSet cur[MT], nxt[MT];
    bf16x8 a0[2][MT], a1[2][MT];
    tile_rows(blk, cur);
    load_chunk(0, cur, a0);

    for (int t = blk; t < ntiles; t += nblk) {
        const int g = t / tiles_per_group;
        if (g != g_acc) { flush_stats(); g_acc = g; }
        tile_rows(t + nblk, nxt);

        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto compute_chunk = [&](int ch, bf16x8 (&a)[2][MT]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ch * 64 + ks * 32 >= K) continue;
                if (p.dbg & 2) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][0][0] += (float)a[ks][mt][0];
                    continue;
                }
                bf16x8 bfr[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bfr[nt] = *reinterpret_cast<const bf16x8*>(Bs + (nt * 16 + fr) * LDB + ch * 64 + ks * 32 + fq * 8);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)       // transposed product: rows = channels, columns = pixel rows
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], a[ks][mt], acc[mt][nt], 0, 0, 0);
            }
        };

        // invariant: a0 holds chunk 0 of this tile (in flight)
        for (int ch = 0; ch < nch; ch += 2) {
            if (ch + 1 < nch) load_chunk(ch + 1, cur, a1);
            compute_chunk(ch, a0);
            if (ch + 1 < nch) {
                if (ch + 2 < nch) load_chunk(ch + 2, cur, a0);
                else load_chunk(0, nxt, a0);              // next tile's first chunk flies under the epilogue
                compute_chunk(ch + 1, a1);
            } else {
                load_chunk(0, nxt, a0);
            }
        }

        // ---- epilogue: lane owns channels nt*16 + fq*4 .. +3 of pixel row mt*16 + fr
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int opix = cur[mt].opix;
            if (opix < 0 || ((p.dbg & 4) && acc[mt][0][0] != 12345.f)) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n0 = nt * 16 + fq * 4;
                if (n0 >= c.N) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j];
                if (p.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += b[j];
                }
                if (DGRAD) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.d_r + (size_t)opix * p.d_ld + n0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 tb = tabs[g * BN + n0 + j];
                        const float rr = (float)rv[j];
                        const float x = v[j] * act_bwd(p.d_act, rr * tb.x + tb.y);
                        v[j] = x;
                        if (want_red) { s1[nt][j] += x; s2[nt][j] += x * (rr - tb.z) * tb.w; }
                    }
                } else if (want_stats) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s1[nt][j] += v[j]; s2[nt][j] += v[j] * v[j]; }
                }
                if (p.out_bf) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
                    *reinterpret_cast<bf16x4*>(p.out_bf + (size_t)opix * p.ldo + n0) = o;
                }
                if (!DGRAD && p.out_act_bf) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16)act_fwd(p.e_act, v[j]);
                    *reinterpret_cast<bf16x4*>(p.out_act_bf + (size_t)opix * p.ldo + n0) = o;
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) cur[mt] = nxt[mt];
    }
    flush_stats();
}

template <int NT, int MT, int NW, bool DGRAD>
int launch_direct(GemmParams p, const double* work, hipStream_t stream) {
    static const int dbg = getenv("MMVAE_DIRECT_DBG") ? atoi(getenv("MMVAE_DIRECT_DBG")) : 0;
    p.dbg = dbg;
    constexpr int BN = NT * 16, BMD = NW * MT * 16;
    const GatherCommon& c = p.c;
    int max_kpad = 0;
    for (int i = 0; i < c.nclasses; ++i) max_kpad = max(max_kpad, p.cls[i].Kpad);
    const size_t lds = (size_t)BN * (max_kpad + 8) * sizeof(bf16) + 32 * sizeof(int) + (DGRAD ? (size_t)c.groups * BN * sizeof(float4) : 0);
    static const int cap = getenv("MMVAE_DIRECT_WGS") ? atoi(getenv("MMVAE_DIRECT_WGS")) : 4;
    const int per_cu = (int)max((size_t)1, min((size_t)cap, (size_t)(160 * 1024) / (lds + 512)));
    const int budget = 256 * per_cu;         // resident workgroups of the whole chip
    double total = 0;
    for (int i = 0; i < c.nclasses; ++i) total += work[i];
    // workgroups per class in proportion to its work, never more than its tiles
    p.dblk[0] = 0;
    for (int i = 0; i < c.nclasses; ++i) {
        const int tiles = ceil_div(p.cls[i].rows_per_group, BMD) * c.groups;
        int nb = (int)(budget * work[i] / total + 0.5);
        nb = max(1, min(nb, tiles));
        p.dblk[i + 1] = p.dblk[i] + nb;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_direct_kernel<NT, MT, NW, DGRAD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_direct_kernel<NT, MT, NW, DGRAD>), dim3(p.dblk[c.nclasses]), dim3(NW * 64), lds, stream, p);
    return mmvae_check_launch("gemm_direct");
}

template <int NT, bool DGRAD>
int launch_direct_nt(const GemmParams& p, const double* work, bool tall, hipStream_t stream) {
    return tall ? launch_direct<NT, 4, 4, DGRAD>(p, work, stream) : launch_direct<NT, 2, 4, DGRAD>(p, work, stream);
}

}  // namespace

// returns 1 when the problem was launched here, 0 when the caller should use the generic kernels, < 0 on error
int try_launch_gemm_direct(const GemmParams& p, hipStream_t stream) {
    const GatherCommon& c = p.c;
    if (c.N > 64 || c.N % 8 != 0 || p.ldo % 8 != 0 || p.ksplit > 1) return 0;
    if (p.out_f || p.e_mask || p.d_mask || p.d_bcast_n > 0 || p.d_colsum || p.d_cmod > 0) return 0;
    if (p.d_r && (p.d_ld % 8 != 0 || p.out_act_bf || p.colstats)) return 0;
    if (p.d_red && (!p.d_meanrstd || !p.d_r)) return 0;
    const int NT = c.N <= 16 ? 1 : c.N <= 32 ? 2 : 4;
    const int BN = NT * 16;
    int max_kpad = 0;
    double work[MMVAE_MAX_CLASSES];
    long long rows_total = 0, out_elems = 0;
    for (int i = 0; i < c.nclasses; ++i) {
        const GatherClass& k = p.cls[i];
        if (k.TH * k.TW > 1 && (c.C < 32 || (c.C & (c.C - 1)) != 0)) return 0;   // the tap walk must be wave-uniform
        if (k.TH * k.TW > 32) return 0;                                     // tap-validity bit mask
        max_kpad = max(max_kpad, k.Kpad);
        work[i] = (double)k.rows_per_group * c.groups * (k.K + 96);
        rows_total += (long long)k.rows_per_group * c.groups;
    }
    const long long nimg_a = c.a_bcast_n > 0 ? c.a_bcast_n : (long long)c.groups * c.group_n;
    if (nimg_a * c.AH * c.AW * c.Ald * 2 >= (1ll << 31)) return 0;         // 32-bit buffer offsets
    out_elems = (long long)c.groups * c.group_n * c.OH * c.OW;
    if (out_elems >= (1ll << 31) || rows_total >= (1ll << 23)) return 0;
    if ((size_t)BN * (max_kpad + 8) * sizeof(bf16) > (size_t)104 * 1024) return 0;     // weights must stay resident in LDS
    if (rows_total < 16384) return 0;                                                 // small grids: generic / row-tile kernels
    // tile height by how many tiles the chip can be given: 64 rows per wave when that still yields >= ~2 per CU
    const bool tall = rows_total >= (long long)256 * 2 * 256;
    const bool dgrad = p.d_r != nullptr;
    int rc;
    if (NT == 1) rc = dgrad ? launch_direct_nt<1, true>(p, work, tall, stream) : launch_direct_nt<1, false>(p, work, tall, stream);
    else if (NT == 2) rc = dgrad ? launch_direct_nt<2, true>(p, work, tall, stream) : launch_direct_nt<2, false>(p, work, tall, stream);
    else rc = dgrad ? launch_direct_nt<4, true>(p, work, tall, stream) : launch_direct_nt<4, false>(p, work, tall, stream);
    return rc == MMVAE_OK ? 1 : rc;
}
