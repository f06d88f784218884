// Weight-streaming GEMM of the persistent row-block kernels (coco_text_bf16.hip, mlp_tail.hip): one 8-wave workgroup owns
// 16 rows; out[16][N] = A[16][K] (bf16, LDS) x W^T with W streamed from L2 through a register ring.
#pragma once
#include "common.h"

namespace mmvae_sg {

constexpr int NW = 8, NTHR = NW * 64;       // waves / threads of a workgroup

// Weight streaming.  Every wave of the workgroup walks a FIXED list of weight chunks per recurrence step: for each GEMM of
// the step, for each of its MAXT column tiles (wave, wave + 8, ...; a wave with fewer real tiles walks a clamped dummy one
// so that all waves consume the same number of chunks), NCH chunks of up to KCH = 10 k-steps (one 16-byte load per lane
// per k-step).  The chunk counts are compile-time constants and a step's total is a multiple of the ring depth D, so the
// ring slot of every chunk is static: while a chunk feeds the MFMAs, the chunk D positions further down the list -- of
// this tile, the next tile, or the NEXT GEMM, also across the barriers between the GEMMs -- is already in flight.  D
// chunks of 10 KB per wave, 8 waves: 160-240 KB in flight per CU, which is what it takes to stream the 1.24 MB of weights
// a step needs from L2 at bandwidth instead of at one round trip per tile.
constexpr int KCH = 10;
template <int KS, int NT> struct WMat {                     // [16*NT][32*KS] bf16, fragment-major (PackDesc::frag)
    __amdgpu_buffer_rsrc_t r;
    static constexpr int ks = KS, nt = NT, kpad = KS * 32, nch = (KS + KCH - 1) / KCH;
    __device__ explicit WMat(const bf16* w) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(w), 0, NT * 16 * KS * 32 * 2, 0x00020000)) {}
};

// chunk `pos` of this wave's list for matrix m: tile slot pos / nch, k-chunk pos % nch.  pos is a compile-time constant at
// every call site (unrolled loops), so the k-step count and the load offsets are immediates; a wave whose tile slot is past
// the matrix (dummy tile) requests nothing.
template <class M>
__device__ __forceinline__ void load_chunk(bf16x8 (&dst)[KCH], const M& m, int pos, int wave, int lane) {
    const int i = pos / M::nch, c = pos - i * M::nch;
    const int nt = wave + NW * i;
    if (nt >= M::nt) return;
    // fragment-major: one k-step = 1 KB, lane-ordered.  Buffer loads: descriptor + wave-uniform offset in SGPRs, the lane
    // offset is the only VGPR address of the whole weight stream
    const int ub = (nt * M::ks + c * KCH) * 1024;
    const int kc = min(KCH, M::ks - c * KCH);
#pragma unroll
    for (int s = 0; s < KCH; ++s)
        if (s < kc) dst[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(m.r, lane * 16, ub + s * 1024, 0));
}

// out[16][ldo] (fp32, LDS) = A[16][32*ks] (bf16, LDS) * W^T (+ cinit, a [rows][ldc] fp32 matrix in global memory whose row
// r0.. block is the time-invariant part of the projection: it enters as the accumulators' initial value).  MAXT = tile
// slots per wave of this GEMM (>= ceil(nt/8); more pads the step's chunk count to a multiple of D), mn = the next GEMM of
// the list (its first chunks are requested from here), SLOT0 = the ring slot of this GEMM's first chunk.
template <int MAXT, int D, int SLOT0, bool CINIT = false, class M, class MN>
__device__ __forceinline__ void stream_gemm(const bf16* A, int lda, const M& m, float* out, int ldo, bf16x8 (&ring)[D][KCH],
                                            const MN& mn, bool has_next, int wave, int lane, const float* cinit = nullptr,
                                            int ldc = 0, int ncols = 0, int rows_ok = 0) {
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int NCH = M::nch, NQ = NCH * MAXT;
    static_assert(MAXT * NW >= M::nt, "tile slots");
    f32x4 ci[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        ci[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (CINIT) {            // rows / columns past the matrix read a clamped element: they land in rows / pad columns nobody reads
            const int col = min((wave + NW * i) * 16 + fr, ncols - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) ci[i][j] = cinit[(size_t)min(fq * 4 + j, rows_ok - 1) * ldc + col];
        }
    }
    bf16x8 af[NCH][KCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int s = 0; s < KCH; ++s)
            if (c * KCH + s < M::ks) af[c][s] = *reinterpret_cast<const bf16x8*>(A + fr * lda + (c * KCH + s) * 32 + fq * 8);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int nt = wave + NW * i;
        f32x4 acc = ci[i];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int q = i * NCH + c;
            const int slot = (SLOT0 + q) % D;
            if (nt < M::nt) {
#pragma unroll
                for (int s = 0; s < KCH; ++s)
                    if (c * KCH + s < M::ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c][s], ring[slot][s], acc, 0, 0, 0);
            }
            // refill this slot with the chunk D positions further down the list
            if (q + D < NQ) load_chunk(ring[slot], m, q + D, wave, lane);
            else if (has_next) load_chunk(ring[slot], mn, q + D - NQ, wave, lane);
        }
        if (nt < M::nt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(fq * 4 + j) * ldo + nt * 16 + fr] = acc[j];
        }
    }
}


}  // namespace mmvae_sg
