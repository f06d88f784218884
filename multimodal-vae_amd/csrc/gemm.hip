// MFMA implicit-GEMM kernels for gfx950 (wave64, v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// gemm_gather_kernel<NT>: block = 4 waves stacked along M (BM=128), BN = 16*NT, BK = 64.
//   Operand A is gathered pixel rows of an NHWC bf16 tensor (register staged so the BatchNorm affine +
//   Swish/ReLU (+dropout keep-mask) of the PRODUCER layer is applied while staging: activated tensors are
//   never materialised in HBM).  Operand B is a pre-packed bf16 weight matrix [Npad][Kpad].
//   LDS rows are padded by one 16-B slot (stride 144 B) -> conflict-free ds_read_b128 fragments.
//   Epilogue options: bias, bf16/fp32 store to a strided pixel grid, per-(group,column) BatchNorm statistics,
//   and the "d-activation" form used by backward (multiply by act'(.) of the consumer-side saved tensor and
//   reduce the two BatchNorm-backward sums).
// wgrad_kernel: dW[N][K] += P^T G over pixel rows, both operands transposed on the fly with
//   ds_read_b64_tr_b16 (LDS row stride 160 B makes the 8-row transposed reads conflict-free),
//   split over row chunks with fp32 atomics into a packed gradient matrix.
#include "gemm.h"
#include "gemm_epi.h"
#include "convres.h"
#include "wgrad_ring.h"
#include <cstdlib>

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int LDA = BK + 8;    // bf16 elements per LDS row (144 B)

typedef __attribute__((ext_vector_type(4))) int i32x4g;

template <int NT>
__global__ __launch_bounds__(256) void gemm_gather_kernel(const GemmParams p) {
    constexpr int BN = NT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                 // [2][BM][LDA]
    bf16* Bs = As + 2 * BM * LDA;                             // [2][BN][LDA]

    const GatherCommon& c = p.c;
    const GatherClass& k = p.cls[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_per_group = (k.rows_per_group + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_per_group * c.groups) return;
    const int g = blockIdx.x / tiles_per_group;
    const int row0 = (blockIdx.x - g * tiles_per_group) * BM;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int nblk = blockIdx.y / ksplit, ksi = blockIdx.y - nblk * ksplit;
    const int n0 = nblk * BN;
    const int K = k.K;
    const int nk_all = (K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * ksi / ksplit), kt1 = (int)((long long)nk_all * (ksi + 1) / ksplit);
    const int nk = kt1 - kt0;
    const int pix_per_img = k.OY * k.OX;

    // ---- per-thread staging coordinates: 4 A rows, vector column kv ----
    const int kv = tid & 7;
    RowCoord rc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = row0 + (tid >> 3) + 32 * i;
        if (r < k.rows_per_group) {
            int rem, ox;
            const int img = fast_divmod(r, pix_per_img, 1.0f / (float)pix_per_img, rem);
            const int oy = fast_divmod(rem, k.OX, 1.0f / (float)k.OX, ox);
            int aimg = g * c.group_n + img;
            if (c.a_bcast_n > 0) aimg %= c.a_bcast_n;
            rc[i].pix = aimg * c.AH * c.AW;
            rc[i].y = oy * c.sy + k.offy;
            rc[i].x = ox * c.sx + k.offx;
        } else {
            rc[i].pix = -1; rc[i].y = 0; rc[i].x = 0;
        }
    }

    // Operand A goes through a buffer descriptor: 32-bit byte offsets, and an invalid vector (padding tap, row past
    // the end, k past K) is an offset of 0xFFFFFFFF that the hardware range check turns into zeros -- no clamped
    // address, no select, no validity state.  The tap walk of this thread's k-vector is kept incrementally (the two
    // integer divisions per k-tile of the first version cost more VALU time than the tile's MFMAs for NT <= 4).
    const int nimg_a = c.a_bcast_n > 0 ? c.a_bcast_n : c.groups * c.group_n;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(c.A), 0, (int)((size_t)nimg_a * c.AH * c.AW * c.Ald * sizeof(bf16)), 0x00020000);
    int rbase[4];                                  // byte offset of (tap 0, channel 0) of this thread's rows
#pragma unroll
    for (int i = 0; i < 4; ++i) rbase[i] = ((rc[i].pix + rc[i].y * c.AW + rc[i].x) * c.Ald) * (int)sizeof(bf16);
    int t_ty, t_tx, t_ac;
    {
        const int kk = kt0 * BK + kv * 8;
        const int tap = kk / c.C;
        t_ac = kk - tap * c.C;
        t_ty = tap / k.TW;
        t_tx = tap - t_ty * k.TW;
    }
    bf16x8 areg[4];
    constexpr int NB = (BN * 8 + 255) / 256;       // B vectors per thread
    bf16x8 breg[NB];
    const bf16* bsrc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int v = (tid + 256 * i) % (BN * 8);
        bsrc[i] = k.Wp + (size_t)(n0 + (v >> 3)) * k.Kpad + (size_t)kt0 * BK + (v & 7) * 8;
    }

    // loads tile kt (tiles are visited in order: the tap walk advances by one k-tile per call)
    auto load_tile = [&](int kt) {
        const bool kin = kt * BK + kv * 8 < K;
        const int toff = (((t_ty * c.dy) * c.AW + t_tx * c.dx) * c.Ald + t_ac) * (int)sizeof(bf16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = rc[i].y + t_ty * c.dy, x = rc[i].x + t_tx * c.dx;
            const bool ok = kin && rc[i].pix >= 0 && (unsigned)y < (unsigned)c.AH && (unsigned)x < (unsigned)c.AW;
            const i32x4g v = __builtin_bit_cast(i32x4g, __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? rbase[i] + toff : -1, 0, 0));
            areg[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            breg[i] = *reinterpret_cast<const bf16x8*>(bsrc[i]);
            bsrc[i] += BK;
        }
        t_ac += BK;
        while (t_ac >= c.C) {
            t_ac -= c.C;
            if (++t_tx == k.TW) { t_tx = 0; ++t_ty; }
        }
    };
    auto store_tile = [&](int buf) {
        bf16* a_dst = As + buf * BM * LDA;
        bf16* b_dst = Bs + buf * BN * LDA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<bf16x8*>(a_dst + ((tid >> 3) + 32 * i) * LDA + kv * 8) = areg[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int v = tid + 256 * i;
            if (v < BN * 8) *reinterpret_cast<bf16x8*>(b_dst + (v >> 3) * LDA + (v & 7) * 8) = breg[i];
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(kt0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt0 + kt + 1);
        const bf16* a_src = As + buf * BM * LDA + (wave * 32 + fr) * LDA + fq * 8;
        const bf16* b_src = Bs + buf * BN * LDA + fr * LDA + fq * 8;
        bf16x8 af[2][2], bfr[2][NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            af[ks][0] = *reinterpret_cast<const bf16x8*>(a_src + ks * 32);
            af[ks][1] = *reinterpret_cast<const bf16x8*>(a_src + 16 * LDA + ks * 32);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[ks][nt] = *reinterpret_cast<const bf16x8*>(b_src + nt * 16 * LDA + ks * 32);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][0], bfr[ks][nt], acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][1], bfr[ks][nt], acc[1][nt], 0, 0, 0);
            }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // The accumulator tile goes through LDS (fp32 [BM][BN+4]; the staging buffers are free after the last barrier)
    // so that every global access is a 16-byte vector of 8 consecutive columns of one row.
    constexpr int LDC = BN + 4;
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ct[(wave * 32 + mt * 16 + fq * 4 + j) * LDC + nt * 16 + fr] = acc[mt][nt][j];
    __syncthreads();
    if (ksplit > 1) {
        // split-K: this workgroup only publishes its partial tile (plain 16-byte stores into its own slab);
        // splitk_finish_kernel sums the slabs and runs the epilogue after the kernel boundary
        const int tile_id = ((int)blockIdx.z * gridDim.x + blockIdx.x) * (gridDim.y / ksplit) + nblk;
        float* slab = p.sk_buf + ((size_t)tile_id * ksplit + ksi) * BM * BN;
        for (int e = tid; e < BM * BN / 4; e += 256) {
            const int r = e / (BN / 4), q = e - r * (BN / 4);
            *reinterpret_cast<f32x4*>(slab + r * BN + q * 4) = *reinterpret_cast<const f32x4*>(ct + r * LDC + q * 4);
        }
        return;
    }
    gemm_epilogue<NT, BM>(p, k, g, row0, n0, ct, smem, tid);
}

template <int NT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const GemmParams p) {
    // one workgroup per (tile, slice of RS rows): every thread sums ONE 8-column vector over the ksplit slabs
    // (2*ksplit independent 16-byte loads in flight), then the common epilogue runs on the slice
    constexpr int BN = NT * 16;
    constexpr int LDC = BN + 4;
    constexpr int VPR = BN / 8;
    constexpr int RS = 256 / VPR;
    constexpr int SLICES = BM / RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ct = reinterpret_cast<float*>(smem);
    const GatherCommon& c = p.c;
    const GatherClass& k = p.cls[blockIdx.z];
    const int tid = threadIdx.x;
    const int tiles_per_group = (k.rows_per_group + BM - 1) / BM;
    const int tile_x = blockIdx.x / SLICES, slice = blockIdx.x - tile_x * SLICES;
    if (tile_x >= tiles_per_group * c.groups) return;
    const int g = tile_x / tiles_per_group;
    const int row0 = (tile_x - g * tiles_per_group) * BM + slice * RS;
    if (row0 >= k.rows_per_group) return;
    const int nblk = blockIdx.y, n0 = nblk * BN;
    const int tile_id = ((int)blockIdx.z * (gridDim.x / SLICES) + tile_x) * gridDim.y + nblk;
    const float* slab = p.sk_buf + (size_t)tile_id * p.ksplit * BM * BN + (size_t)slice * RS * BN;
    const int rl = tid / VPR, cv = tid - rl * VPR;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < p.ksplit; ++q) {
        const float* s = slab + (size_t)q * BM * BN + rl * BN + cv * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(s);
        const f32x4 b = *reinterpret_cast<const f32x4*>(s + 4);
        lo[0] += a[0]; lo[1] += a[1]; lo[2] += a[2]; lo[3] += a[3];
        hi[0] += b[0]; hi[1] += b[1]; hi[2] += b[2]; hi[3] += b[3];
    }
    *reinterpret_cast<f32x4*>(ct + rl * LDC + cv * 8) = lo;
    *reinterpret_cast<f32x4*>(ct + rl * LDC + cv * 8 + 4) = hi;
    __syncthreads();
    gemm_epilogue<NT, RS>(p, k, g, row0, n0, ct, smem, tid);
}

// ------------------------------------------------------------------------------------------------
constexpr int WM = 64;          // rows per iteration

template <int LDW>
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int mb, int c0, int lane) {
    // MFMA operand fragment whose k index runs over LDS rows (pixel rows), element i over columns.
    const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row = mb + (g4 >> 1) * 16 + (g4 & 1) * 4 + q;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const bf16* a0 = tile + row * LDW + c0 + 4 * pp;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * LDW));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// Block tile TN x TK = (16*NWT*WN) x (16*KWT*WK), 4 waves as WN x WK, 64 pixel rows per iteration.
// Thread t stages row (t>>2) of the iteration: TN/32 vectors of the plain operand and TK/32 vectors of the gathered
// operand (vector index (t&3) + 4*i), so one row decomposition serves all of a thread's loads; the (tap, channel)
// of each gathered vector is fixed for the whole kernel.  Loads of iteration i+1 are issued before the MFMAs of
// iteration i (register prefetch), LDS is single-buffered.
//
// Row chunks: the pixel rows are split over `chunks` workgroups per output tile.  With a slab (p.slab != nullptr) every
// workgroup stores its partial tile with plain stores into its own slab copy and wgrad_reduce_kernel sums the copies
// (deterministic, and plain stores run at ~5x the chip-wide float-atomic rate: MI355X_MICROARCH "Global float atomics"),
// so the row chunks can be short (many workgroups, several per CU: the loop is a latency chain of gathered loads).
// Without a slab the partial tiles are added into the zeroed packed gradient with fp32 atomics (first version).
template <int NWT, int KWT, int WN, int WK>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const int bx, const int by, const int bz, char* smem) {
    constexpr int TN = 16 * NWT * WN, TK = 16 * KWT * WK;
    constexpr int LDP = TN + 16, LDG = TK + 16;          // +16 elements: conflict-free 8-row transposed reads
    constexpr int NPV = TN / 32, NGV = TK / 32;
    bf16* Ps = reinterpret_cast<bf16*>(smem);            // [64][LDP]
    bf16* Gs = Ps + WM * LDP;                            // [64][LDG]
    const GatherCommon& c = p.c;
    const int ncls = c.nclasses;
    const int cls_i = bz % ncls;
    const int chunk = bz / ncls;
    const GatherClass& k = p.cls[cls_i];
    const int n0 = bx * TN, k0 = by * TK;
    if (k0 >= k.K) return;
    const int rows_total = c.groups * k.rows_per_group;
    const int r_begin = chunk * p.rows_per_block;
    const int r_end = min(rows_total, r_begin + p.rows_per_block);
    if (r_begin >= rows_total && p.slab == nullptr) return;      // (a slab copy must be written even when it is all zeros)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WK, wk = wave % WK;
    const int pix_per_img = k.OY * k.OX;
    const float inv_rpg = 1.0f / (float)k.rows_per_group, inv_pix = 1.0f / (float)pix_per_img, inv_ox = 1.0f / (float)k.OX;
    const int row_l = tid >> 2, sub = tid & 3;

    int g_dy[NGV], g_dx[NGV], g_c[NGV];
    unsigned g_kin = 0;
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
        const int kk = k0 + (sub + 4 * i) * 8;
        g_dy[i] = 0; g_dx[i] = 0; g_c[i] = 0;
        if (kk < k.K) {
            int tap = kk / c.C;
            g_c[i] = kk - tap * c.C;
            int ty = tap / k.TW;
            g_dy[i] = ty * c.dy; g_dx[i] = (tap - ty * k.TW) * c.dx;
            g_kin |= 1u << i;
        }
    }

    f32x4 acc[NWT][KWT];
#pragma unroll
    for (int a = 0; a < NWT; ++a)
#pragma unroll
        for (int b = 0; b < KWT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 pv[NPV], gv[NGV];
    unsigned pok = 0, gok = 0;
    auto load_rows = [&](int rb) {
        const int rq = rb + row_l;
        const bool rin = rq < r_end;
        const int r = rin ? rq : r_end - 1;                  // clamped: loads are unconditional, results masked
        int rg, rem, ox;                                     // reciprocal divisions: this runs once per 64-row iteration
        const int g = fast_divmod(r, k.rows_per_group, inv_rpg, rg);
        const int img = fast_divmod(rg, pix_per_img, inv_pix, rem);
        const int oy = fast_divmod(rem, k.OX, inv_ox, ox);
        const int n = g * c.group_n + img;
        const int an = c.a_bcast_n > 0 ? n % c.a_bcast_n : n;
        const size_t ppix = (size_t)(n * c.OH + oy * c.osy + k.ooy) * c.OW + ox * c.osx + k.oox;
        pok = 0; gok = 0;
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            const int pcol = n0 + (sub + 4 * i) * 8;
            const bool ok = rin && pcol < c.N;               // ldp >= round_up(N,8)
            pv[i] = *reinterpret_cast<const bf16x8*>(p.P + ppix * p.ldp + (pcol < c.N ? pcol : 0));
            pok |= (ok ? 1u : 0u) << i;
        }
        const int y0 = oy * c.sy + k.offy, x0 = ox * c.sx + k.offx;
#pragma unroll
        for (int i = 0; i < NGV; ++i) {
            const int y = y0 + g_dy[i], x = x0 + g_dx[i];
            const bool ok = rin && ((g_kin >> i) & 1) && (unsigned)y < (unsigned)c.AH && (unsigned)x < (unsigned)c.AW;
            const int gp = ok ? (an * c.AH + y) * c.AW + x : 0;
            gv[i] = *reinterpret_cast<const bf16x8*>(c.A + (size_t)gp * c.Ald + g_c[i]);
            gok |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_rows = [&]() {
#pragma unroll
        for (int i = 0; i < NPV; ++i)
            *reinterpret_cast<bf16x8*>(Ps + row_l * LDP + (sub + 4 * i) * 8) = ((pok >> i) & 1) ? pv[i] : zero8();
#pragma unroll
        for (int i = 0; i < NGV; ++i)
            *reinterpret_cast<bf16x8*>(Gs + row_l * LDG + (sub + 4 * i) * 8) = ((gok >> i) & 1) ? gv[i] : zero8();
    };

    if (r_begin < r_end) load_rows(r_begin);
    for (int rb = r_begin; rb < r_end; rb += WM) {
        __syncthreads();                                  // previous iteration's fragment reads are done
        store_rows();
        __syncthreads();
        if (rb + WM < r_end) load_rows(rb + WM);          // prefetch the next 64 rows while the MFMAs run
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[NWT], bfr[KWT];
#pragma unroll
            for (int a = 0; a < NWT; ++a) af[a] = tr_frag<LDP>(Ps, ks * 32, (wn * NWT + a) * 16, lane);
#pragma unroll
            for (int b = 0; b < KWT; ++b) bfr[b] = tr_frag<LDG>(Gs, ks * 32, (wk * KWT + b) * 16, lane);
#pragma unroll
            for (int a = 0; a < NWT; ++a)
#pragma unroll
                for (int b = 0; b < KWT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
    if (p.slab) {
        // partial tile -> this chunk's slab copy of class cls_i ([slab_rows][Kpad], rows n < slab_rows = tiles*TN)
        float* dst = p.slab + (size_t)chunk * p.slab_chunk_stride + p.slab_cls_off[cls_i];
#pragma unroll
        for (int a = 0; a < NWT; ++a)
#pragma unroll
            for (int b = 0; b < KWT; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + (wn * NWT + a) * 16 + fq * 4 + j;
                    const int kc = k0 + (wk * KWT + b) * 16 + fr;
                    if (kc < k.Kpad) dst[(size_t)n * k.Kpad + kc] = acc[a][b][j];
                }
        return;
    }
#pragma unroll
    for (int a = 0; a < NWT; ++a)
#pragma unroll
        for (int b = 0; b < KWT; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + (wn * NWT + a) * 16 + fq * 4 + j;
                const int kc = k0 + (wk * KWT + b) * 16 + fr;
                if (n < c.N && kc < k.K) atomicAdd(k.dWp + (size_t)n * k.Kpad + kc, acc[a][b][j]);
            }
}

template <int NWT, int KWT, int WN, int WK>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_body<NWT, KWT, WN, WK>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Several independent weight-gradient problems in ONE launch (the GRU / Linear layers of the text path: ~0.1 GFLOP each,
// every one of them a launch at the kernel-latency floor before): blockIdx.z runs over (problem, class x chunk), the x/y
// extent is the largest tile grid of the group (the others return early).
constexpr int WGRAD_MULTI_MAX = 6;
struct WgradMulti {
    WgradParams p[WGRAD_MULTI_MAX];
    int zoff[WGRAD_MULTI_MAX + 1];       // prefix sums of the z extents
    int gx[WGRAD_MULTI_MAX], gy[WGRAD_MULTI_MAX];
    int n;
};
static_assert(sizeof(WgradMulti) <= 3800, "kernel arguments are limited to 4 KB");
template <int NWT, int KWT, int WN, int WK>
__global__ __launch_bounds__(256) void wgrad_multi_kernel(const WgradMulti m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.z >= m.zoff[i + 1]) ++i;
    if ((int)blockIdx.x >= m.gx[i] || (int)blockIdx.y >= m.gy[i]) return;
    wgrad_body<NWT, KWT, WN, WK>(m.p[i], blockIdx.x, blockIdx.y, blockIdx.z - m.zoff[i], smem);
}

// dst[n][k] += sum over chunks of slab[chunk][n][k]  for n < N, k < K.  One thread per (4 consecutive k, group of up to
// REDUCE_CG chunks): a job with many chunks and few outputs (the thin first / last conv layers: 512 outputs, ~1000 chunks)
// is spread over chunk groups whose partial sums meet in dst through float atomics (a few thousand atomics in all);
// jobs with <= REDUCE_CG chunks -- every MFMA-sized layer -- are summed in a fixed order with a plain read-modify-write.
constexpr int REDUCE_MAX_JOBS = 40;
constexpr int REDUCE_CG = 32;
struct WgradReduceArgs {
    struct Job { float* dst; const float* slab; int N, K, Kpad, chunks; long long chunk_stride; int first_block, groups, src_ld; } job[REDUCE_MAX_JOBS];
    int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceArgs a) {
    int j = 0;
    while (j + 1 < a.n && (int)blockIdx.x >= a.job[j + 1].first_block) ++j;
    const WgradReduceArgs::Job& q = a.job[j];
    const int kv = q.Kpad / 4;
    const long long t = (long long)(blockIdx.x - q.first_block) * 256 + threadIdx.x;
    const long long nvec = (long long)q.N * kv;
    if (t >= nvec * q.groups) return;
    const int grp = (int)(t / nvec);
    const long long e = t - (long long)grp * nvec;
    const int n = (int)(e / kv), k4 = (int)(e - (long long)n * kv) * 4;
    if (k4 >= q.K) return;
    const int c0 = grp * REDUCE_CG, c1 = min(q.chunks, c0 + REDUCE_CG);
    const float* src = q.slab + (size_t)n * q.src_ld + k4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int cidx = c0;
    for (; cidx + 8 <= c1; cidx += 8) {                        // 8 independent 16-byte loads in flight
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(cidx + i) * q.chunk_stride);
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; cidx < c1; ++cidx) s += *reinterpret_cast<const f32x4*>(src + (size_t)cidx * q.chunk_stride);
    float* d = q.dst + (size_t)n * q.Kpad + k4;
    if (q.groups == 1) {
        f32x4 o = *reinterpret_cast<f32x4*>(d);
        o += s;
        *reinterpret_cast<f32x4*>(d) = o;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(d + i, s[i]);
    }
}

struct WgradCfg { int TN, TK; };
inline WgradCfg wgrad_cfg(const WgradParams& p) {
    int max_k = 0;
    for (int i = 0; i < p.c.nclasses; ++i) max_k = max(max_k, p.cls[i].K);
    // tile shape by problem shape: thin outputs get a wide K tile so each staged pixel row feeds more MFMAs
    if (max_k <= 64) return {32, 64};
    if (p.c.N <= 32) return {32, 256};
    if (p.c.N <= 64) return {64, 256};
    return {128, 128};
}

// grid / row-chunk / slab set-up shared by the single and the grouped launch
int wgrad_setup(WgradParams& p, WgradSlabCtx* ctx, dim3& grid, size_t& lds, hipStream_t stream, int group_size = 1) {
    const WgradCfg cfg = wgrad_cfg(p);
    const int TN = cfg.TN, TK = cfg.TK;
    const GatherCommon& c = p.c;
    int max_rows = 0, max_k = 0;
    for (int i = 0; i < c.nclasses; ++i) {
        max_rows = max(max_rows, c.groups * p.cls[i].rows_per_group);
        max_k = max(max_k, p.cls[i].K);
    }
    const int tiles = ceil_div(c.N, TN) * ceil_div(max_k, TK) * c.nclasses;
    constexpr int env_it = 0, blk_target = 384, slab_target = 1024, slab_it = 2;      // tuned on MultiMNIST / CelebA
    p.slab = nullptr;
    int chunks;
    // slab form: ~4 workgroups per CU, each at least 2 iterations long (a slab copy costs one plain-store pass and one
    // read in the reduce kernel, not a float-atomic pass)
    long long slab_elems = 0;
    for (int i = 0; i < c.nclasses; ++i) slab_elems += (long long)ceil_div(c.N, TN) * TN * p.cls[i].Kpad;
    if (ctx && ctx->pool) {
        // (a grouped launch fills the chip with its problems together; every slab copy is written once and read once,
        //  so a launch's copies are capped at SLAB_CAP floats: the big-weight layers have few rows per output anyway)
        constexpr long long slab_cap = 3ll << 20;
        chunks = max(1, min(ceil_div(max_rows, slab_it * WM), ceil_div(max(64, slab_target / group_size), tiles)));
        chunks = (int)min((long long)chunks, max(1ll, slab_cap / slab_elems));
        while (chunks > 1 && ctx->used + (size_t)chunks * slab_elems > ctx->cap) chunks = (chunks + 1) / 2;
        if (chunks > 1 && ctx->used + (size_t)chunks * slab_elems <= ctx->cap && (int)ctx->jobs.size() + c.nclasses <= REDUCE_MAX_JOBS) {
            p.slab = ctx->pool + ctx->used;
            p.slab_chunk_stride = slab_elems;
            long long off = 0;
            p.rows_per_block = round_up(ceil_div(max_rows, chunks), WM);
            const int real_chunks = ceil_div(max_rows, p.rows_per_block);
            for (int i = 0; i < c.nclasses; ++i) {
                p.slab_cls_off[i] = off;
                WgradSlabJob j{};
                j.dst = p.cls[i].dWp; j.slab = p.slab + off; j.N = c.N; j.K = p.cls[i].K; j.Kpad = p.cls[i].Kpad;
                j.chunks = real_chunks; j.chunk_stride = slab_elems; j.stream = stream;
                ctx->jobs.push_back(j);
                off += (long long)ceil_div(c.N, TN) * TN * p.cls[i].Kpad;
            }
            ctx->used += (size_t)real_chunks * slab_elems;
            chunks = real_chunks;
        } else {
            chunks = 0;      // decided below
        }
    } else {
        chunks = 0;
    }
    if (!p.slab) {
        // atomic form: every workgroup adds its whole tile into the packed gradient with fp32 atomics, so fewer, longer
        // workgroups win once there are enough of them (measured at B=256: >= 8 iterations of 64 rows and 128..384
        // workgroups: step 0.991 -> 0.965 ms against 4 iterations / 768 workgroups); small problems keep the shorter chain
        const int min_it = env_it > 0 ? env_it : (max_rows >= 2048 ? 8 : 4);
        chunks = max(1, min(ceil_div(max_rows, min_it * WM), ceil_div(blk_target, tiles)));
        p.rows_per_block = round_up(ceil_div(max_rows, chunks), WM);
        chunks = ceil_div(max_rows, p.rows_per_block);
    }
    grid = dim3(ceil_div(c.N, TN), ceil_div(max_k, TK), chunks * c.nclasses);
    lds = (size_t)WM * (TN + 16 + TK + 16) * sizeof(bf16);
    return MMVAE_OK;
}

template <int NWT, int KWT, int WN, int WK>
int launch_wgrad_v(const WgradParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<NWT, KWT, WN, WK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    }
    MMVAE_LAUNCH((wgrad_kernel<NWT, KWT, WN, WK>), grid, dim3(256), lds, stream, p);
    return mmvae_check_launch("wgrad");
}

// ------------------------------------------------------------------------------------------------
// gemm_rowtile_kernel: the small-M regime (a few hundred..thousand pixel rows, long K: classifier / upsample
// Linears, the 2x2 bottleneck convs and their data gradients).  A 128-row tile would leave most CUs idle behind a
// serial chain of K-tile latencies; here one workgroup owns 16 rows and ALL output columns: the gathered 16 x K
// operand sits in LDS, every wave streams its share of the packed weight rows straight from L2 into MFMA B
// fragments (independent 16-byte loads, deep in flight), so the chip runs rows/16 workgroups with no inter-tile
// dependency.  Same GemmParams / epilogue semantics as gemm_gather_kernel.
constexpr int RT_R = 16;
constexpr int RT_KC = 2048;            // K chunk staged in LDS

template <int NTW>                      // N tiles (of 16 columns) per wave, 4 waves: N <= 64*NTW
__global__ __launch_bounds__(256) void gemm_rowtile_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    struct RowInfo { int apix, y, x, ok; long long opix, rpix, grow; };
    __shared__ RowInfo rinfo[RT_R];
    const GatherCommon& c = p.c;
    const GatherClass& k = p.cls[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_per_group = (k.rows_per_group + RT_R - 1) / RT_R;
    if ((int)blockIdx.x >= tiles_per_group * c.groups) return;
    const int g = blockIdx.x / tiles_per_group;
    const int row0 = (blockIdx.x - g * tiles_per_group) * RT_R;
    const int K = k.K, N = c.N;
    const int kc_max = min(round_up(K, 32), RT_KC);
    const int LDK = kc_max + 8;
    bf16* As = reinterpret_cast<bf16*>(smem);                          // [16][LDK]
    const int LDC = round_up(N, 16) + 4;
    float* ct = reinterpret_cast<float*>(As + RT_R * LDK);            // [16][LDC]
    float2* cstat = reinterpret_cast<float2*>(ct + RT_R * LDC);       // [round_up(N,16)] column sums
    const int pix_per_img = k.OY * k.OX;
    if (tid < RT_R) {
        RowInfo ri{};
        const int r = row0 + tid;
        ri.ok = r < k.rows_per_group;
        if (ri.ok) {
            const int img = r / pix_per_img, rem = r - img * pix_per_img;
            const int oy = rem / k.OX, ox = rem - oy * k.OX;
            const int nimg = g * c.group_n + img;
            const int aimg = c.a_bcast_n > 0 ? nimg % c.a_bcast_n : nimg;
            ri.apix = aimg * c.AH * c.AW; ri.y = oy * c.sy + k.offy; ri.x = ox * c.sx + k.offx;
            ri.opix = (long long)(nimg * c.OH + oy * c.osy + k.ooy) * c.OW + ox * c.osx + k.oox;
            const int rimg = p.d_bcast_n > 0 ? nimg % p.d_bcast_n : nimg;
            ri.rpix = (long long)(rimg * c.OH + oy * c.osy + k.ooy) * c.OW + ox * c.osx + k.oox;
            ri.grow = (long long)g * k.rows_per_group + r;
        }
        rinfo[tid] = ri;
    }
    for (int i = tid; i < round_up(N, 16); i += 256) cstat[i] = make_float2(0.f, 0.f);
    const int ntiles = (N + 15) / 16;

    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc0 = 0; kc0 < K; kc0 += RT_KC) {
        const int kc_len = min(round_up(K - kc0, 32), RT_KC);
        __syncthreads();                                               // rinfo ready / previous chunk consumed
        // ---- stage the gathered 16 x kc_len operand chunk (unconditional clamped loads, zero-fill)
        const int vpr = kc_len / 8;
        for (int v = tid; v < RT_R * vpr; v += 256) {
            const int row = v / vpr, kk = kc0 + (v - row * vpr) * 8;
            const RowInfo ri = rinfo[row];
            bool ok = ri.ok && kk < K;
            int tap = 0, ch = 0, ty = 0, tx = 0;
            if (kk < K) { tap = kk / c.C; ch = kk - tap * c.C; ty = tap / k.TW; tx = tap - ty * k.TW; }
            const int y = ri.y + ty * c.dy, x = ri.x + tx * c.dx;
            ok = ok && (unsigned)y < (unsigned)c.AH && (unsigned)x < (unsigned)c.AW;
            const int pix = ok ? ri.apix + y * c.AW + x : 0;
            bf16x8 val = *reinterpret_cast<const bf16x8*>(c.A + (size_t)pix * c.Ald + ch);
            if (!ok) val = zero8();
            *reinterpret_cast<bf16x8*>(As + row * LDK + (kk - kc0)) = val;
        }
        __syncthreads();
        // ---- MFMA: this wave's column tiles over the whole chunk; B fragments come straight from L2.
        // All loads of a k-group (KG k-steps x NTW tiles) are issued unconditionally (clamped) before its MFMAs,
        // so each wave keeps NTW*KG independent 16-byte loads in flight instead of one dependent chain per tile.
        constexpr int KG = NTW <= 4 ? 4 : (NTW <= 8 ? 2 : 1);
        const int ksteps = kc_len / 32;
        const bf16* a_src = As + fr * LDK + fq * 8;
        const bf16* wrow[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int nt = min(t * 4 + wave, ntiles - 1);
            wrow[t] = k.Wp + (size_t)(nt * 16 + fr) * k.Kpad + kc0 + fq * 8;
        }
        for (int ks0 = 0; ks0 < ksteps; ks0 += KG) {
            bf16x8 bq[NTW][KG], aq[KG];
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                const int ks = min(ks0 + j, ksteps - 1);
#pragma unroll
                for (int t = 0; t < NTW; ++t) bq[t][j] = *reinterpret_cast<const bf16x8*>(wrow[t] + ks * 32);
                aq[j] = *reinterpret_cast<const bf16x8*>(a_src + ks * 32);
            }
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                if (ks0 + j < ksteps) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[j], bq[t][j], acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- accumulators -> LDS tile
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int nt = t * 4 + wave;
        if (nt >= ntiles) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) ct[(fq * 4 + j) * LDC + nt * 16 + fr] = acc[t][j];
    }
    __syncthreads();
    // ---- epilogue: 16-byte vectors of 8 columns (same semantics as gemm_gather_kernel's)
    const bool want_stats = p.colstats != nullptr;
    const bool want_red = p.d_red != nullptr || p.d_colsum != nullptr;
    const int nvec = (N + 7) / 8;
    const bool ld_ok = (p.ldo % 8 == 0);
    for (int v = tid; v < RT_R * nvec; v += 256) {
        const int row = v / nvec, col0 = (v - row * nvec) * 8;
        const RowInfo ri = rinfo[row];
        if (!ri.ok) continue;
        const bool vec_ok = ld_ok && col0 + 8 <= N;
        float val[8], rr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = min(col0 + j, N - 1);
            val[j] = ct[row * LDC + col0 + j] + (p.bias ? p.bias[col] : 0.f);
        }
        if (p.d_r) {
            if (vec_ok && p.d_ld % 8 == 0) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(p.d_r + ri.rpix * p.d_ld + col0);
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = (float)rv[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = (col0 + j < N) ? (float)p.d_r[ri.rpix * p.d_ld + col0 + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = min(col0 + j, N - 1);
                float sc = 1.f, sh = 0.f;
                if (p.d_affine) { const float2 a = p.d_affine[g * N + col]; sc = a.x; sh = a.y; }
                float x = val[j] * act_bwd(p.d_act, rr[j] * sc + sh);
                if (p.d_mask) x = (col0 + j < N && p.d_mask[ri.grow * N + col0 + j]) ? x * p.d_mask_scale : 0.f;
                val[j] = x;
                if (want_red && col0 + j < N) {
                    float2 mr = p.d_meanrstd ? p.d_meanrstd[g * N + col] : make_float2(0.f, 0.f);
                    atomicAdd(&cstat[col0 + j].x, x);
                    atomicAdd(&cstat[col0 + j].y, x * (rr[j] - mr.x) * mr.y);
                }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (col0 + j < N) { atomicAdd(&cstat[col0 + j].x, val[j]); atomicAdd(&cstat[col0 + j].y, val[j] * val[j]); }
        }
        float av[8];
        if (p.out_act_bf) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = act_fwd(p.e_act, val[j]);
                if (p.e_mask) x = (col0 + j < N && p.e_mask[ri.grow * N + col0 + j]) ? x * p.e_mask_scale : 0.f;
                av[j] = x;
            }
        }
        if (vec_ok) {
            if (p.out_bf) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)val[j];
                *reinterpret_cast<bf16x8*>(p.out_bf + ri.opix * p.ldo + col0) = o;
            }
            if (p.out_act_bf) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)av[j];
                *reinterpret_cast<bf16x8*>(p.out_act_bf + ri.opix * p.ldo + col0) = o;
            }
            if (p.out_f) {
                *reinterpret_cast<f32x4*>(p.out_f + ri.opix * p.ldo + col0) = f32x4{val[0], val[1], val[2], val[3]};
                *reinterpret_cast<f32x4*>(p.out_f + ri.opix * p.ldo + col0 + 4) = f32x4{val[4], val[5], val[6], val[7]};
            }
        } else {
            for (int j = 0; j < 8 && col0 + j < N; ++j) {
                if (p.out_bf) p.out_bf[ri.opix * p.ldo + col0 + j] = (bf16)val[j];
                if (p.out_act_bf) p.out_act_bf[ri.opix * p.ldo + col0 + j] = (bf16)av[j];
                if (p.out_f) p.out_f[ri.opix * p.ldo + col0 + j] = val[j];
            }
        }
    }
    if (want_stats || want_red) {
        __syncthreads();
        float2* dst = want_stats ? p.colstats : p.d_red;
        const int slot = (blockIdx.x + 5 * blockIdx.z) % MMVAE_STAT_SLOTS;
        for (int col = tid; col < N; col += 256) {
            const float2 s = cstat[col];
            if (dst) {
                atomicAdd(&dst[(g * MMVAE_STAT_SLOTS + slot) * N + col].x, s.x);
                atomicAdd(&dst[(g * MMVAE_STAT_SLOTS + slot) * N + col].y, s.y);
            }
            if (p.d_colsum) atomicAdd(p.d_colsum + col, s.x);
        }
    }
}

template <int NTW>
int launch_rowtile(const GemmParams& p, hipStream_t stream) {
    int max_tiles = 0, max_k = 0;
    for (int i = 0; i < p.c.nclasses; ++i) {
        max_tiles = max(max_tiles, ceil_div(p.cls[i].rows_per_group, RT_R));
        max_k = max(max_k, p.cls[i].K);
    }
    const int kc = min(round_up(max_k, 32), RT_KC);
    const size_t lds = (size_t)RT_R * (kc + 8) * sizeof(bf16) + (size_t)RT_R * (round_up(p.c.N, 16) + 4) * sizeof(float) +
                       (size_t)round_up(p.c.N, 16) * sizeof(float2);
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rowtile_kernel<NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    }
    dim3 grid(max_tiles * p.c.groups, 1, p.c.nclasses);
    MMVAE_LAUNCH(gemm_rowtile_kernel<NTW>, grid, dim3(256), lds, stream, p);
    return mmvae_check_launch("gemm_rowtile");
}

template <int NT>
int launch_gemm_nt(const GemmParams& p, hipStream_t stream) {
    constexpr int BN = NT * 16;
    int max_tiles = 0;
    for (int i = 0; i < p.c.nclasses; ++i) max_tiles = max(max_tiles, ceil_div(p.cls[i].rows_per_group, BM));
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    dim3 grid(max_tiles * p.c.groups, ceil_div(p.c.N, BN) * ksplit, p.c.nclasses);
    size_t lds = (size_t)2 * (BM + BN) * LDA * sizeof(bf16);
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_gather_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&splitk_finish_kernel<NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    }
    hipEvent_t held = ksplit > 1 ? mmvae_take_stop_event() : nullptr;    // a completion event belongs to the LAST kernel of the launch
    MMVAE_LAUNCH(gemm_gather_kernel<NT>, grid, dim3(256), lds, stream, p);
    MMVAE_TRY(mmvae_check_launch("gemm_gather"));
    if (ksplit > 1) {
        if (held) mmvae_arm_stop_event(held);
        constexpr int SLICES = BM / (256 / (BN / 8));
        dim3 fgrid(max_tiles * p.c.groups * SLICES, ceil_div(p.c.N, BN), p.c.nclasses);
        MMVAE_LAUNCH(splitk_finish_kernel<NT>, fgrid, dim3(256), (size_t)32 * 1024, stream, p);
        return mmvae_check_launch("splitk_finish");
    }
    return MMVAE_OK;
}

}  // namespace

int launch_gemm_gather(const GemmParams& p, hipStream_t stream) {
    const GatherCommon& c = p.c;
    {
        double f = 0;
        for (int i = 0; i < c.nclasses; ++i) f += 2.0 * c.groups * p.cls[i].rows_per_group * (double)c.N * p.cls[i].K;
        mmvae_count_flops(f);
    }
    MMVAE_REQUIRE(c.nclasses >= 1 && c.nclasses <= MMVAE_MAX_CLASSES, "gemm: bad class count %d", c.nclasses);
    MMVAE_REQUIRE(c.C % 8 == 0 && c.Ald % 8 == 0, "gemm: C=%d / Ald=%d must be multiples of 8", c.C, c.Ald);
    MMVAE_REQUIRE(c.groups >= 1 && c.group_n >= 1 && c.N >= 1, "gemm: empty problem");
    for (int i = 0; i < c.nclasses; ++i) {
        const GatherClass& k = p.cls[i];
        MMVAE_REQUIRE(k.Kpad % BK == 0 && k.Kpad >= k.K && k.K == k.TH * k.TW * c.C, "gemm: class %d K=%d Kpad=%d", i, k.K, k.Kpad);
        MMVAE_REQUIRE(k.rows_per_group == c.group_n * k.OY * k.OX && k.rows_per_group > 0, "gemm: class %d rows", i);
        MMVAE_REQUIRE(k.Wp != nullptr, "gemm: class %d has no weights", i);
        MMVAE_REQUIRE((k.OY - 1) * c.osy + k.ooy < c.OH && (k.OX - 1) * c.osx + k.oox < c.OW, "gemm: class %d output grid", i);
    }
    MMVAE_REQUIRE(c.a_mask == nullptr && c.a_affine == nullptr && c.a_act == ACT_NONE,
                  "gemm: operand transforms are not supported (activations are materialised by bn_act / the epilogue)");
    MMVAE_REQUIRE(p.ksplit <= 1 || p.sk_buf != nullptr, "gemm: split-K needs scratch");
    {
        const long long nimg_a = c.a_bcast_n > 0 ? c.a_bcast_n : (long long)c.groups * c.group_n;
        MMVAE_REQUIRE(nimg_a * c.AH * c.AW * c.Ald * 2 < (1ll << 31), "gemm: gathered operand exceeds 2 GiB (32-bit buffer offsets)");
        for (int i = 0; i < c.nclasses; ++i)
            MMVAE_REQUIRE(p.cls[i].rows_per_group < (1 << 23), "gemm: more than 2^23 rows per group");
    }
    {   // conv layers with an image-resident kernel compiled for their geometry (convres.hip)
        const int rc = try_launch_convres(p, stream);
        if (rc != 0) return rc < 0 ? rc : MMVAE_OK;
    }
    if (mmvae_probe_on()) {
        char tag[96];
        double f = 0;
        for (int i = 0; i < c.nclasses; ++i) f += 2.0 * c.groups * p.cls[i].rows_per_group * (double)c.N * p.cls[i].K;
        snprintf(tag, sizeof(tag), "gemm %d>%d %dx%d>%dx%d taps%dx%d %s img%d", c.C, c.N, c.AH, c.AW, c.OH, c.OW, p.cls[0].TH, p.cls[0].TW,
                 p.d_r ? "dgrad" : "fwd", c.groups * c.group_n);
        mmvae_probe_tag(tag, f);
    }
    {
        const int rc = try_launch_gemm_small(p, stream);
        if (rc != 0) return rc < 0 ? rc : MMVAE_OK;
    }
    {
        // few 128-row tiles and a long K loop: use the 16-row weight-streaming kernel (rows/16 workgroups)
        int max_tiles = 0, min_k = 1 << 30, max_k = 0;
        for (int i = 0; i < c.nclasses; ++i) {
            max_tiles = max(max_tiles, ceil_div(p.cls[i].rows_per_group, BM));
            min_k = min(min_k, p.cls[i].K);
            max_k = max(max_k, p.cls[i].K);
        }
        const int tiles128 = max_tiles * c.groups * c.nclasses * ceil_div(c.N, 128);
        // (long K chains are better served by split-K on the 128-row kernel: each 16-row workgroup would have to
        //  stream the whole weight matrix through one CU)
        if (p.ksplit <= 1 && tiles128 <= 96 && c.N <= 1024 && min_k >= 96 && max_k <= 384) {
            if (c.N <= 256) return launch_rowtile<4>(p, stream);
            if (c.N <= 512) return launch_rowtile<8>(p, stream);
            return launch_rowtile<16>(p, stream);
        }
    }
    if (c.N <= 16) return launch_gemm_nt<1>(p, stream);
    if (c.N <= 32) return launch_gemm_nt<2>(p, stream);
    if (c.N <= 64) return launch_gemm_nt<4>(p, stream);
    return launch_gemm_nt<8>(p, stream);
}

static int wgrad_validate(const WgradParams& p) {
    const GatherCommon& c = p.c;
    {
        double f = 0;
        for (int i = 0; i < c.nclasses; ++i) f += 2.0 * c.groups * p.cls[i].rows_per_group * (double)c.N * p.cls[i].K;
        mmvae_count_flops(f);
        if (mmvae_probe_on()) {
            char tag[96];
            snprintf(tag, sizeof(tag), "wgrad %d>%d %dx%d>%dx%d taps%dx%d img%d", c.C, c.N, c.AH, c.AW, c.OH, c.OW, p.cls[0].TH, p.cls[0].TW,
                     c.groups * c.group_n);
            mmvae_probe_tag(tag, f);
        }
    }
    MMVAE_REQUIRE(c.nclasses >= 1 && c.nclasses <= MMVAE_MAX_CLASSES, "wgrad: bad class count %d", c.nclasses);
    MMVAE_REQUIRE(c.C % 8 == 0 && c.Ald % 8 == 0 && p.ldp % 8 == 0 && p.ldp >= round_up(c.N, 8),
                  "wgrad: C/Ald/ldp must be multiples of 8 and ldp >= round_up(N,8)");
    MMVAE_REQUIRE(c.a_mask == nullptr && c.a_affine == nullptr && c.a_act == ACT_NONE && p.p_affine == nullptr && p.p_act == ACT_NONE,
                  "wgrad: operand transforms are not supported");
    for (int i = 0; i < c.nclasses; ++i) {
        const GatherClass& k = p.cls[i];
        MMVAE_REQUIRE(k.Kpad >= k.K && k.Kpad % 4 == 0 && k.K == k.TH * k.TW * c.C, "wgrad: class %d K=%d Kpad=%d", i, k.K, k.Kpad);
        MMVAE_REQUIRE(k.rows_per_group == c.group_n * k.OY * k.OX && k.dWp != nullptr, "wgrad: class %d", i);
        MMVAE_REQUIRE((long long)c.groups * k.rows_per_group < (1 << 23), "wgrad: more than 2^23 rows");
    }
    return MMVAE_OK;
}

int launch_wgrad(const WgradParams& pin, hipStream_t stream, WgradSlabCtx* ctx) {
    const int skip = mmvae_knob("dbg_skip_wgrad", 0);           // measurement aid: the step without (1: all, 2: the ring-staged, 3: the
    if (skip == 1) return MMVAE_OK;                             // streamed single-problem, 4: the grouped) weight gradients
    MMVAE_TRY(wgrad_validate(pin));
    WgradParams p = pin;
    {   // conv layers with a ring-staged kernel compiled for their geometry (wgrad_ring.hip)
        const int rc = try_launch_wgrad_ring(p, stream, ctx);
        if (rc != 0) return rc < 0 ? rc : MMVAE_OK;
    }
    if (skip == 3) return MMVAE_OK;
    dim3 grid; size_t lds;
    MMVAE_TRY(wgrad_setup(p, ctx, grid, lds, stream));
    const WgradCfg cfg = wgrad_cfg(p);
    if (cfg.TK == 64) return launch_wgrad_v<2, 1, 1, 4>(p, grid, lds, stream);          //  32 x  64
    if (cfg.TN == 32) return launch_wgrad_v<2, 4, 1, 4>(p, grid, lds, stream);          //  32 x 256
    if (cfg.TN == 64) return launch_wgrad_v<4, 4, 1, 4>(p, grid, lds, stream);          //  64 x 256
    return launch_wgrad_v<4, 4, 2, 2>(p, grid, lds, stream);                            // 128 x 128
}

// Several problems of the 128 x 128 tile class in one launch; anything that does not fit the grouped kernel (another tile
// shape, more than WGRAD_MULTI_MAX problems) is launched on its own.
int launch_wgrad_group(const WgradParams* list, int n, hipStream_t stream, WgradSlabCtx* ctx) {
    if (mmvae_knob("dbg_skip_wgrad", 0) == 1 || mmvae_knob("dbg_skip_wgrad", 0) == 4) return MMVAE_OK;
    constexpr bool no_group = false;
    int i = 0;
    while (i < n) {
        WgradMulti m{};
        int gxm = 0, gym = 0;
        while (i < n && m.n < WGRAD_MULTI_MAX) {
            MMVAE_TRY(wgrad_validate(list[i]));
            const WgradCfg cfg = wgrad_cfg(list[i]);
            if (cfg.TN != 128 || no_group) {
                if (m.n > 0) break;                       // flush the group first (keeps the issue order)
                MMVAE_TRY(launch_wgrad(list[i], stream, ctx));
                ++i;
                continue;
            }
            WgradParams p = list[i];
            dim3 grid; size_t lds;
            MMVAE_TRY(wgrad_setup(p, ctx, grid, lds, stream, min(n, WGRAD_MULTI_MAX)));
            m.p[m.n] = p; m.gx[m.n] = (int)grid.x; m.gy[m.n] = (int)grid.y;
            m.zoff[m.n + 1] = m.zoff[m.n] + (int)grid.z;
            gxm = max(gxm, (int)grid.x); gym = max(gym, (int)grid.y);
            ++m.n; ++i;
        }
        if (m.n == 0) continue;
        static std::atomic<unsigned> attr_set{0};
        if (mmvae_first_use_on_device(attr_set))
            hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_multi_kernel<4, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        const size_t lds = (size_t)WM * (128 + 16 + 128 + 16) * sizeof(bf16);
        MMVAE_LAUNCH((wgrad_multi_kernel<4, 4, 2, 2>), dim3(gxm, gym, m.zoff[m.n]), dim3(256), lds, stream, m);
        MMVAE_TRY(mmvae_check_launch("wgrad_multi"));
    }
    return MMVAE_OK;
}

int launch_wgrad_reduce(WgradSlabCtx* ctx, hipStream_t stream, bool only_own) {
    MMVAE_TRY(launch_wgrad_ring_reduce(ctx, stream, only_own));
    if (!ctx || ctx->jobs.empty()) return MMVAE_OK;
    std::vector<WgradSlabJob> todo, keep;
    for (const WgradSlabJob& j : ctx->jobs) (only_own && j.stream != stream ? keep : todo).push_back(j);
    ctx->jobs.swap(keep);
    size_t done = 0;
    while (done < todo.size()) {
        WgradReduceArgs a{};
        int blocks = 0;
        while (done < todo.size() && a.n < REDUCE_MAX_JOBS) {
            const WgradSlabJob& j = todo[done++];
            WgradReduceArgs::Job& q = a.job[a.n++];
            q.dst = j.dst; q.slab = j.slab; q.N = j.N; q.K = j.K; q.Kpad = j.Kpad; q.chunks = j.chunks; q.chunk_stride = j.chunk_stride;
            q.first_block = blocks;
            q.src_ld = j.src_ld > 0 ? j.src_ld : j.Kpad;
            q.groups = ceil_div(j.chunks, REDUCE_CG);
            blocks += (int)(((long long)j.N * (j.Kpad / 4) * q.groups + 255) / 256);
        }
        MMVAE_LAUNCH(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
        MMVAE_TRY(mmvae_check_launch("wgrad_reduce"));
    }
    return MMVAE_OK;
}
