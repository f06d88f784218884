// The two small dependent Linears at the end of the image encoder's classifier (multimnist/model.py:173-179:
// Linear(400,200) -> Swish -> Dropout -> Linear(200,2D)) as ONE row-block launch, and their data gradients as one more.
//
// As separate launches they were 9.8 + 6.4 us forward and 15.2 + 17.1 us backward plus a launch gap each, all on the
// step's critical chain, for 0.1 GFLOP.  Rows are independent, so a workgroup takes 16 rows through both layers: the
// intermediate stays in LDS, the weights (266 / 272 KB bf16, fragment-major) stream from L2 through the register ring of
// stream_gemm.h, and the epilogues (bias, Swish, dropout, raw + activated copies for the backward pass; d-Swish, dropout,
// bias-gradient column sums) are those of the GEMM epilogue they replace, value for value.
#include "mlp_tail.h"
#include "stream_gemm.h"

namespace {

using namespace mmvae_sg;

constexpr int TR = 16;
constexpr int N1 = 400, N2 = 200, N3 = 200;                        // fc1 / fc2 / fc3 widths (n_latents = 100)
constexpr int K1P = 416, K2P = 224, K3P = 224;                     // padded reduction widths (multiples of 32)
constexpr int LDX1 = K1P + 8, LDX2 = K2P + 8;                      // bf16 LDS row strides (+16 B: conflict-free ds_read_b128)
constexpr int LDO = 404;                                           // fp32 LDS row stride of a GEMM result (<= 400 columns)
constexpr int NQ2 = (N2 + 31) / 32, NQ1 = (N1 + 31) / 32;          // 7, 13

__global__ __launch_bounds__(NTHR) void mlp2_fwd_kernel(const Mlp2FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* o = reinterpret_cast<float*>(smem);                      // [16][LDO]
    bf16* xb = reinterpret_cast<bf16*>(o + TR * LDO);               // [16][LDX1] fc2 operand (activated fc1 output)
    bf16* hb = xb + TR * LDX1;                                      // [16][LDX2] fc3 operand (activated fc2 output)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * TR, rows = a.rows;
    const int grow = tid >> 5, c0 = tid & 31;
    const bool gok = r0 + grow < rows;
    const size_t gr = gok ? r0 + grow : 0;
    constexpr int D = 4;             // the whole of classifier.3's chunk list is requested before the first MFMA
    const WMat<K1P / 32, (N2 + 15) / 16> w2(a.w2);
    const WMat<K2P / 32, (N3 + 15) / 16> w3(a.w3);
    bf16x8 ring[D][KCH];
#pragma unroll
    for (int q = 0; q < D; ++q) load_chunk(ring[q], w2, q, wave, lane);
    // operand rows -> LDS (16-byte vectors; pad columns and rows past the end are zero)
    for (int i = tid; i < TR * (LDX1 / 8); i += NTHR) {
        const int row = i / (LDX1 / 8), v = i - row * (LDX1 / 8);
        bf16x8 x = {};
        if (r0 + row < rows && v * 8 < N1) x = *reinterpret_cast<const bf16x8*>(a.x + (size_t)(r0 + row) * N1 + v * 8);
        *reinterpret_cast<bf16x8*>(xb + row * LDX1 + v * 8) = x;
    }
    for (int i = tid; i < TR * LDX2; i += NTHR) hb[i] = (bf16)0.f;
    __syncthreads();
    // ---- fc2: 13 column tiles over 8 waves (2 slots), 13 k-steps in 2 chunks = 4 chunks per wave
    uint8_t kp[NQ2];                 // (requested before the GEMM: nothing of the epilogue waits on memory)
    float bb[NQ2];
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
        const int j = min(c0 + 32 * q, N2 - 1);
        kp[q] = a.mask ? a.mask[gr * N2 + j] : (uint8_t)1;
        bb[q] = a.b2[j];
    }
    stream_gemm<2, D, 0>(xb, LDX1, w2, o, LDO, ring, w3, true, wave, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
        const int j = c0 + 32 * q;
        if (j < N2) {
            const float v = o[grow * LDO + j] + bb[q];
            float x = act_fwd(ACT_SWISH, v);
            if (a.mask) x = kp[q] ? x * a.mask_scale : 0.f;
            hb[grow * LDX2 + j] = (bf16)x;
            if (gok) { a.y2[gr * N2 + j] = (bf16)v; a.ay2[gr * N2 + j] = (bf16)x; }
        }
    }
    __syncthreads();
    // ---- fc3: 13 tiles, 7 k-steps = 2 chunks per wave
    stream_gemm<2, D, 0>(hb, LDX2, w3, o, LDO, ring, w3, false, wave, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
        const int j = c0 + 32 * q;
        if (j < N3 && gok) a.out[gr * N3 + j] = o[grow * LDO + j] + a.b3[j];
    }
}

// d_y2 = (d_out W3) * swish'(y2) * keep2 ;  d_y1 = (d_y2 W2) * swish'(y1) * keep1 ;  bias gradients = column sums
__global__ __launch_bounds__(NTHR) void mlp2_bwd_kernel(const Mlp2BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* o = reinterpret_cast<float*>(smem);                      // [16][LDO]
    float* cs = o + TR * LDO;                                       // [NQ1][16][32] column-sum staging (q, row lane, column lane)
    bf16* db = reinterpret_cast<bf16*>(cs + NQ1 * TR * 32);         // [16][LDX2] d_out, then d_y2
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * TR, rows = a.rows;
    const int grow = tid >> 5, c0 = tid & 31;
    const bool gok = r0 + grow < rows;
    const size_t gr = gok ? r0 + grow : 0;
    constexpr int D = 4;
    const WMat<K3P / 32, (N2 + 15) / 16> w3t(a.w3t);               // [208][224]: rows = fc2 units, k = fc3 outputs
    const WMat<K2P / 32, N1 / 16> w2t(a.w2t);                      // [400][224]: rows = fc1 units, k = fc2 units
    bf16x8 ring[D][KCH];
    load_chunk(ring[0], w3t, 0, wave, lane); load_chunk(ring[1], w3t, 1, wave, lane);       // the 2 chunks of the first GEMM
    load_chunk(ring[2], w2t, 0, wave, lane); load_chunk(ring[3], w2t, 1, wave, lane);       // and the first 2 of the second
    for (int i = tid; i < TR * (LDX2 / 8); i += NTHR) {
        const int row = i / (LDX2 / 8), v = i - row * (LDX2 / 8);
        bf16x8 x = {};
        if (r0 + row < rows && v * 8 < N3) x = *reinterpret_cast<const bf16x8*>(a.d_out + (size_t)(r0 + row) * N3 + v * 8);
        *reinterpret_cast<bf16x8*>(db + row * LDX2 + v * 8) = x;
    }
    __syncthreads();
    // ---- through fc3: 13 tiles x 7 k-steps = 2 chunks per wave
    bf16 rr2[NQ2], rr1[NQ1];         // raw pre-activations and keep flags of both layers: requested up front, so that
    uint8_t kp2[NQ2], kp1[NQ1];      // neither epilogue waits on memory
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
        const int j = min(c0 + 32 * q, N2 - 1);
        rr2[q] = a.y2[gr * N2 + j];
        kp2[q] = a.mask2 ? a.mask2[gr * N2 + j] : (uint8_t)1;
    }
#pragma unroll
    for (int q = 0; q < NQ1; ++q) {
        const int j = min(c0 + 32 * q, N1 - 1);
        rr1[q] = a.y1[gr * N1 + j];
        kp1[q] = a.mask1 ? a.mask1[gr * N1 + j] : (uint8_t)1;
    }
    stream_gemm<2, D, 0>(db, LDX2, w3t, o, LDO, ring, w2t, true, wave, lane);
    __syncthreads();
    // bias gradients = column sums: a thread owns columns c0 + 32 q of ONE row; the 16 rows of a column meet in LDS
    auto colsums = [&](int ncols, float* dst) {
        __syncthreads();
        for (int col = tid; col < ncols; col += NTHR) {
            const float* p = cs + (col >> 5) * TR * 32 + (col & 31);
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < TR; ++r) t += p[r * 32];
            atomicAdd(dst + col, t);
        }
    };
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
        const int j = c0 + 32 * q;
        float x = 0.f;
        if (j < N2 && gok) {
            x = o[grow * LDO + j] * act_bwd(ACT_SWISH, (float)rr2[q]);
            if (a.mask2) x = kp2[q] ? x * a.mask_scale : 0.f;
            a.dy2[gr * N2 + j] = (bf16)x;
        }
        cs[(q * TR + grow) * 32 + c0] = x;
        if (j < N2) db[grow * LDX2 + j] = (bf16)x;       // d_out is consumed (barrier behind the GEMM): the operand of the next one
    }
    colsums(N2, a.db2);
    __syncthreads();
    // ---- through fc2: 25 tiles (4 slots; the last holds one real tile) x 7 k-steps = 4 chunks per wave
    stream_gemm<4, D, 2>(db, LDX2, w2t, o, LDO, ring, w2t, false, wave, lane);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ1; ++q) {
        const int j = c0 + 32 * q;
        float x = 0.f;
        if (j < N1 && gok) {
            x = o[grow * LDO + j] * act_bwd(ACT_SWISH, (float)rr1[q]);
            if (a.mask1) x = kp1[q] ? x * a.mask_scale : 0.f;
            a.dy1[gr * N1 + j] = (bf16)x;
        }
        cs[(q * TR + grow) * 32 + c0] = x;
    }
    colsums(N1, a.db1);
}

}  // namespace

int launch_mlp2_fwd(const Mlp2FwdArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.rows > 0 && a.x && a.w2 && a.w3 && a.b2 && a.b3 && a.y2 && a.ay2 && a.out, "mlp2_fwd: null argument");
    const size_t lds = (size_t)TR * LDO * sizeof(float) + (size_t)TR * (LDX1 + LDX2) * sizeof(bf16);
    MMVAE_LAUNCH(mlp2_fwd_kernel, dim3(ceil_div(a.rows, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("mlp2_fwd");
}
int launch_mlp2_bwd(const Mlp2BwdArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.rows > 0 && a.d_out && a.w3t && a.w2t && a.y2 && a.y1 && a.dy2 && a.dy1 && a.db2 && a.db1, "mlp2_bwd: null argument");
    const size_t lds = (size_t)(TR * LDO + NQ1 * TR * 32) * sizeof(float) + (size_t)TR * LDX2 * sizeof(bf16);
    MMVAE_LAUNCH(mlp2_bwd_kernel, dim3(ceil_div(a.rows, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("mlp2_bwd");
}
