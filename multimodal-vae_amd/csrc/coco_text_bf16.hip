// Caption recurrences of the COCO MMVAE (coco/model.py:219-312) as persistent launches: decoder forward, decoder BPTT,
// encoder forward, encoder BPTT -- four launches where the fp32 path (coco_text.hip) issues 3 dependent launches per forward
// step and 5 per backward step of each recurrence (17 ms per training step at the per-GPU batch of configuration 5, all of it
// dependent-launch latency).
//
//   * single-workgroup form (coco_dec_fwd/bwd_kernel): one 8-wave workgroup owns 16 rows of the 3B-row decoder batch for the
//     WHOLE recurrence; state lives in LDS / registers; the weights (1.34 MB bf16 per step forward, 1.26 MB backward) stream
//     from L2 through a register ring (stream_gemm.h) as fragment-major 16-byte loads, ACROSS the barriers between the GEMMs
//     of a step: a step costs the L2 stream of one CU (24 us), not a latency chain.  Used when the row blocks fill the chip.
//   * cluster form (coco_dec_fwd/bwd_cl_kernel): P = 8 or 4 workgroups per row block, each owning a slice of the hidden
//     units / embedding columns of every product, three all-gathers per step through global memory (12-13 us per step).
//   * caption encoder (coco_enc_fwd/bwd_res_kernel): W_hh resident in registers + LDS, gate math in the MFMA accumulators;
//     the streamed form (coco_enc_fwd/bwd_kernel) is kept for batches that are not a multiple of 4.
//   * bf16 MFMA operands, fp32 accumulation, gate math and state in fp32.  Measured against the reference (oracle with the
//     GEMM operands of the caption GRUs rounded to bf16, 102 steps, B=16): losses move by <= 1e-4 relative, gradient
//     tensors by <= 3e-3 -- inside the 1e-3 ELBO bound, so the 1/16-rate fp32 MFMA is not needed.
// The weight gradients are batched GEMMs over all T*R rows afterwards (operands saved as bf16 in [t][row] layout).
#include "coco_plan.h"
#include "stream_gemm.h"

namespace {

using namespace mmvae_sg;

constexpr int TR = 16;
constexpr int H = COCO_H, G = COCO_G, E = COCO_E;
constexpr int HP = CTB_HP, XP = CTB_XP, GP = CTB_GP, EP = CTB_EP;   // 224, 320, 608, 304
constexpr int LDH = HP + 8, LDX = XP + 8, LDGK = GP + 8;            // bf16 LDS row strides (+16 B: conflict-free ds_read_b128)
constexpr int LDG = 612, LDO = 308, LDT = 212;                      // fp32 LDS row strides

__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }

// The element-wise phases between the GEMMs map thread -> (row = tid / 32, columns c0 + 32 q): every global address of a
// phase is one base plus immediates, and the phase's global loads are all issued before the first one is used.
constexpr int NQH = (H + 31) / 32;      // 7
constexpr int NQE = (E + 31) / 32;      // 10

// ================================================================== forward
template <bool KEEP, bool SAVE>
__global__ __launch_bounds__(NTHR) void coco_dec_fwd_kernel(const CocoDecFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ga = reinterpret_cast<float*>(smem);                     // [16][LDG] input projection / output projection
    float* gb = ga + TR * LDG;                                      // [16][LDG] hidden projection
    float* h0f = gb + TR * LDG;                                     // [16][H]
    float* h1f = h0f + TR * H;
    float* bias = h1f + TR * H;                                     // b_hh0 | b_ih1 | b_hh1, [3][G]
    bf16* xb = reinterpret_cast<bf16*>(bias + 3 * G);               // [16][LDX] current input vector
    bf16* h0b = xb + TR * LDX;                                      // [16][LDH]
    bf16* midb = h0b + TR * LDH;
    bf16* h1b = midb + TR * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR
    const int r0 = blockIdx.x * TR, R = a.R, T = a.T;
    const size_t RH = (size_t)R * H;
    const int grow = tid >> 5, c0 = tid & 31;
    const bool gok = r0 + grow < R;
    const int rows_ok = min(TR, R - r0);
    const size_t gr = gok ? r0 + grow : 0;

    // ---- state: h0 = h1 = z2h(z) (given), x = '<s>'; pad columns of the bf16 operands are zero for the whole kernel
    for (int i = tid; i < TR * LDX; i += NTHR) xb[i] = (bf16)0.f;
    for (int i = tid; i < 3 * TR * LDH; i += NTHR) h0b[i] = (bf16)0.f;
    for (int i = tid; i < G; i += NTHR) { bias[i] = a.bhh0[i]; bias[G + i] = a.bih1[i]; bias[2 * G + i] = a.bhh1[i]; }
    __syncthreads();
    for (int i = tid; i < TR * H; i += NTHR) {
        const int row = i / H, j = i - row * H;
        const float v = r0 + row < R ? a.hinit[(size_t)(r0 + row) * H + j] : 0.f;
        h0f[i] = v; h1f[i] = v;
        h0b[row * LDH + j] = (bf16)v; h1b[row * LDH + j] = (bf16)v;
        if (SAVE && r0 + row < R) {
            a.h0b_all[(size_t)(r0 + row) * HP + j] = (bf16)v; a.h1b_all[(size_t)(r0 + row) * HP + j] = (bf16)v;
        }
    }
    // column H of the saved hidden-state operands is 1.0: the batched weight gradients then carry the bias gradients in it
    if (SAVE && c0 == 0 && gok) { a.h0b_all[gr * HP + H] = (bf16)1.f; a.h1b_all[gr * HP + H] = (bf16)1.f; }
    for (int i = tid; i < TR * E; i += NTHR) {
        const int row = i / E, e = i - row * E;
        const bf16 v = (bf16)a.sos[e];
        xb[row * LDX + e] = v;
        if (SAVE && r0 + row < R) a.xb_all[(size_t)(r0 + row) * XP + e] = v;
    }
    // chunk schedule of a step (ring depth 3): ih0 5, hh0 5, ih1 5, hh1 5, ho 4 (3 real tile slots + 1 dummy) = 24 chunks
    constexpr int D = 3;
    const WMat<XP / 32, GP / 16> d_ih0(a.w_ih0);
    const WMat<HP / 32, GP / 16> d_hh0(a.w_hh0), d_ih1(a.w_ih1), d_hh1(a.w_hh1);
    const WMat<HP / 32, EP / 16> d_ho(a.w_ho);
    bf16x8 ring[D][KCH];
#pragma unroll
    for (int q = 0; q < D; ++q) load_chunk(ring[q], d_ih0, q, wave, lane);
    __syncthreads();
    const float* zi0 = a.zi0 + (size_t)r0 * G;
    const float* zo = a.zo + (size_t)r0 * E;
    const float *pa = ga + grow * LDG, *pb = gb + grow * LDG;

    for (int t = 0; t < T; ++t) {
        // ---- layer 0: both projections (the z-part of the input projection enters as the accumulators' initial value)
        stream_gemm<5, D, 0, true>(xb, LDX, d_ih0, ga, LDG, ring, d_hh0, true, wave, lane, zi0, G, G, rows_ok);
        stream_gemm<5, D, 2>(h0b, LDH, d_hh0, gb, LDG, ring, d_ih1, true, wave, lane);
        uint8_t kp[NQH];
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = c0 + 32 * q;
            kp[q] = 1;
            if (KEEP) kp[q] = gok && j < H ? a.keep[(size_t)t * RH + gr * H + j] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = c0 + 32 * q;
            if (j < H) {
                const int i = grow * H + j;
                const float r = sigm(pa[j] + pb[j] + bias[j]);
                const float z = sigm(pa[H + j] + pb[H + j] + bias[H + j]);
                const float ghn = pb[2 * H + j] + bias[2 * H + j];
                const float n = tanh_fast(pa[2 * H + j] + r * ghn);
                const float hn = (1.0f - z) * n + z * h0f[i];
                h0f[i] = hn;
                h0b[grow * LDH + j] = (bf16)hn;
                float mid = hn;
                if (KEEP) mid = kp[q] ? hn * a.keep_scale : 0.f;
                midb[grow * LDH + j] = (bf16)mid;
                if (SAVE && gok) {
                    a.h0_all[(size_t)(t + 1) * RH + gr * H + j] = hn;
                    float* s = a.sav0 + ((size_t)t * R + gr) * 4 * H;
                    s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
                    a.h0b_all[((size_t)(t + 1) * R + gr) * HP + j] = (bf16)hn;
                    a.midb_all[((size_t)t * R + gr) * HP + j] = (bf16)mid;
                }
            }
        }
        if (SAVE && c0 == 0 && gok) {
            a.h0b_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
            a.midb_all[((size_t)t * R + gr) * HP + H] = (bf16)1.f;
            a.h1b_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
        }
        __syncthreads();
        // ---- layer 1
        stream_gemm<5, D, 1>(midb, LDH, d_ih1, ga, LDG, ring, d_hh1, true, wave, lane);
        stream_gemm<5, D, 0>(h1b, LDH, d_hh1, gb, LDG, ring, d_ho, true, wave, lane);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = c0 + 32 * q;
            if (j < H) {
                const int i = grow * H + j;
                const float r = sigm(pa[j] + bias[G + j] + pb[j] + bias[2 * G + j]);
                const float z = sigm(pa[H + j] + bias[G + H + j] + pb[H + j] + bias[2 * G + H + j]);
                const float ghn = pb[2 * H + j] + bias[2 * G + 2 * H + j];
                const float n = tanh_fast(pa[2 * H + j] + bias[G + 2 * H + j] + r * ghn);
                const float hn = (1.0f - z) * n + z * h1f[i];
                h1f[i] = hn;
                h1b[grow * LDH + j] = (bf16)hn;
                if (SAVE && gok) {
                    a.h1_all[(size_t)(t + 1) * RH + gr * H + j] = hn;
                    float* s = a.sav1 + ((size_t)t * R + gr) * 4 * H;
                    s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
                    a.h1b_all[((size_t)(t + 1) * R + gr) * HP + j] = (bf16)hn;
                }
            }
        }
        __syncthreads();
        // ---- output projection: the word vector of this step, fed back as the next input (coco/model.py:284-286)
        const bool last = t + 1 == T;
        stream_gemm<4, D, 2, true>(h1b, LDH, d_ho, ga, LDG, ring, d_ih0, !last, wave, lane, zo, E, E, rows_ok);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQE; ++q) {
            const int e = c0 + 32 * q;
            if (e < E) {
                const float v = pa[e];
                xb[grow * LDX + e] = (bf16)v;
                if (gok) {
                    a.sentence[(gr * T + t) * E + e] = v;
                    if (SAVE && !last) a.xb_all[((size_t)(t + 1) * R + gr) * XP + e] = (bf16)v;
                }
            }
        }
        __syncthreads();
    }
}

// ================================================================== forward, CLUSTER form
// At the per-GPU batch of configuration 5 (R = 384 rows = 24 row blocks) a decoder step is bound by what ONE CU can pull out
// of L2 (1.34 MB at ~55 GB/s = 24 us) while 230 CUs idle.  Here P workgroups (ranks) share a row block: rank r owns the
// hidden-unit blocks r, r + P, ... of EVERY gate GEMM (weights packed per gate, so r, z and n of a unit stay together) and
// the embedding tiles r, r + P, ... of the output projection, i.e. streams 1/P of the weights, computes the new state of
// its units, and the ranks all-gather the new state after each of the three phases of a step through global memory.
// Exchange (guide Guideline 16, recipe R1): every rank writes its slice of a phase buffer with write-through (sc1) stores,
// drains them (s_waitcnt vmcnt(0), workgroup barrier), and ONE lane stores the rank's flag = epoch; a reader polls the P
// flags (one lane each), passes a barrier, and reads the whole buffer with 16-byte sc1 loads (every load of handed-off
// bytes is sc1: no acquire fence).  Epoch = 3 t + phase + 1, one buffer per phase: a rank can be at most one phase ahead
// of the slowest one; buffers and flags are zeroed before every launch; every spin is bounded (timeout word: the step's
// results are then garbage, loudly, instead of a hang).  The first version exchanged 8-byte {epoch, 2 x bf16} granules
// with device-scope atomics (the data is the flag, one hop): correct, but one uncached load per granule and reader --
// 15,200 per BPTT step -- made the exchange, not the weight stream, the bound.
// Results equal the single-workgroup kernels up to the fp32 summation order of split reductions.
constexpr unsigned CL_SPIN_MAX = 1u << 22;

struct ClX {
    __amdgpu_buffer_rsrc_t rs;      // this cluster's phase buffer
    unsigned* flags;                // [P] epochs
};
__device__ __forceinline__ ClX cl_x(char* base, int bytes, unsigned* flags) {
    return ClX{__builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000), flags};
}
__device__ __forceinline__ void cl_signal(const ClX& x, int rank, unsigned epoch, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave: its payload stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_store(x.flags + rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int P>
__device__ __forceinline__ void cl_wait(const ClX& x, unsigned epoch, unsigned* tmo, int tid) {
    if (tid < P) {
        unsigned spin = 0;
        while (__hip_atomic_load(x.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            if (++spin >= CL_SPIN_MAX) { atomicExch(tmo, epoch); break; }       // gave up: the step's results are garbage, the word says so
            if ((spin & 7) == 7) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the sc1 loads below the poll)
}
__device__ __forceinline__ bf16x8 cl_load16(const ClX& x, int byte_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(x.rs, byte_off, 0, 16));     // aux 16 = sc1
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4c;
__device__ __forceinline__ void cl_store4(const ClX& x, int byte_off, bf16 a, bf16 b) {
    const unsigned v = ((unsigned)__builtin_bit_cast(unsigned short, b) << 16) | __builtin_bit_cast(unsigned short, a);
    __builtin_amdgcn_raw_buffer_store_b32(v, x.rs, byte_off, 0, 16);
}
__device__ __forceinline__ void cl_store8(const ClX& x, int byte_off, bf16 a, bf16 b, bf16 c, bf16 d) {
    u32x2 v;
    v[0] = ((unsigned)__builtin_bit_cast(unsigned short, b) << 16) | __builtin_bit_cast(unsigned short, a);
    v[1] = ((unsigned)__builtin_bit_cast(unsigned short, d) << 16) | __builtin_bit_cast(unsigned short, c);
    __builtin_amdgcn_raw_buffer_store_b64(v, x.rs, byte_off, 0, 16);
}
constexpr int CLF_A = TR * H * 4, CLF_B = TR * H * 2, CLF_C = TR * E * 2;        // forward phase buffers (bytes)
constexpr int CLF_BYTES = CLF_A + CLF_B + CLF_C + 256;                         // + flags [3][P <= 8] (padded)
constexpr int CLB_G = TR * H * 8, CLB_C = TR * E * 2;                          // BPTT: gate gradients (x2), output gradient
constexpr int CLB_BYTES = 2 * CLB_G + CLB_C + 256;

struct ClMat { __amdgpu_buffer_rsrc_t r; int ks; };     // gate matrices: per-gate packs, tile g * 13 + ub; ho: tile et

// chunk `pos` (= tile slot of this wave; one chunk per tile, every K here is <= 10 k-steps) of the rank's tile list
template <bool GATES, int P>
__device__ __forceinline__ void cl_load(bf16x8 (&dst)[KCH], const ClMat& m, int pos, int nown, int rank, int wave, int lane) {
    const int li = wave + NW * pos;
    if (li >= nown) return;
    const int tile = GATES ? (li % 3) * 13 + rank + P * (li / 3) : rank + P * li;
    const int ub = tile * m.ks * 1024;
#pragma unroll
    for (int s = 0; s < KCH; ++s)
        if (s < m.ks) dst[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(m.r, lane * 16, ub + s * 1024, 0));
}

// out[16][ldo], columns li*16 .. of this wave's tiles, = A * W^T (+ cinit).  KS = k-steps of THIS matrix.
template <int KS, int MAXT, int MAXTN, int D, int SLOT0, bool GATES, bool GATESN, int P, bool CINIT>
__device__ __forceinline__ void cl_gemm(const bf16* A, int lda, const ClMat& m, int nown, float* out, int ldo, bf16x8 (&ring)[D][KCH],
                                        const ClMat& mn, int nown_n, bool has_next, int rank, int wave, int lane,
                                        const float* cinit, int ldc, int climit, int gstride, int rows_ok) {
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 ci[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        ci[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (CINIT) {
            const int li = wave + NW * i;
            const int within = (GATES ? rank + P * (li / 3) : rank + P * li) * 16 + fr;     // unit / embedding column
            const bool ok = li < nown && within < climit;
            const int col = ok ? (GATES ? (li % 3) * gstride + within : within) : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) ci[i][j] = cinit[(size_t)min(fq * 4 + j, rows_ok - 1) * ldc + col];
        }
    }
    bf16x8 af[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const bf16x8*>(A + fr * lda + s * 32 + fq * 8);
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int li = wave + NW * i;
        const int slot = (SLOT0 + i) % D;
        f32x4 acc = ci[i];
        if (li < nown) {
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], ring[slot][s], acc, 0, 0, 0);
        }
        if (i + D < MAXT) cl_load<GATES, P>(ring[slot], m, i + D, nown, rank, wave, lane);
        else if (has_next && i + D - MAXT < MAXTN) cl_load<GATESN, P>(ring[slot], mn, i + D - MAXT, nown_n, rank, wave, lane);
        if (li < nown) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(fq * 4 + j) * ldo + li * 16 + fr] = acc[j];
        }
    }
}

template <bool KEEP, bool SAVE, int P>
__global__ __launch_bounds__(NTHR) void coco_dec_fwd_cl_kernel(const CocoDecFwdArgs a) {
    constexpr int NUBMAX = (13 + P - 1) / P, NOEMAX = (19 + P - 1) / P;           // unit blocks / embedding tiles per rank (max)
    constexpr int MT = (3 * NUBMAX + NW - 1) / NW, MTO = (NOEMAX + NW - 1) / NW;    // tile slots per wave
    constexpr int D = MT;                                                          // ring depth = one gate GEMM ahead
    static_assert(MTO <= MT, "output projection slots");
    constexpr int LDC = 3 * NUBMAX * 16 + 4;                                       // fp32 LDS row stride of the GEMM results
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ga = reinterpret_cast<float*>(smem);                     // [16][LDC]
    float* gb = ga + TR * LDC;
    float* bias = gb + TR * LDC;                                    // b_hh0 | b_ih1 | b_hh1, [3][G]
    bf16* xb = reinterpret_cast<bf16*>(bias + 3 * G);               // [16][LDX]
    bf16* h0b = xb + TR * LDX;
    bf16* midb = h0b + TR * LDH;
    bf16* h1b = midb + TR * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (a.R + TR - 1) / TR, nblk_pad = (nblk + 7) / 8 * 8;           // ranks of a cluster: equal blockIdx % 8 (one XCD
    const int rank = blockIdx.x / nblk_pad, blk = blockIdx.x - rank * nblk_pad;    //  under round-robin placement; speed only)
    if (blk >= nblk) return;
    const int r0 = blk * TR, R = a.R, T = a.T;
    const size_t RH = (size_t)R * H;
    const int rows_ok = min(TR, R - r0);
    const int nub = (13 - rank + P - 1) / P, noe = (19 - rank + P - 1) / P;        // owned unit blocks / embedding tiles
    const int ng = 3 * nub;
    const int grow = tid >> 5, pl = tid & 31;
    const bool gok = r0 + grow < R;
    const size_t gr = gok ? r0 + grow : 0;
    // this thread's pair of owned units (gate phases): local units 2*pl, 2*pl + 1 of the rank's nub*16
    const int uk = (2 * pl) >> 4, uu = (2 * pl) & 15;
    const int j0 = (rank + P * uk) * 16 + uu;
    const bool uok = uk < nub && j0 < H;
    char* xbase = reinterpret_cast<char*>(a.cl_xchg) + (size_t)blk * CLF_BYTES;     // phase buffers + flags of this cluster
    unsigned* xflags = reinterpret_cast<unsigned*>(xbase + CLF_A + CLF_B + CLF_C);
    const ClX xA = cl_x(xbase, CLF_A, xflags);                      // [16][200] {h0, dropout(h0)}
    const ClX xB = cl_x(xbase + CLF_A, CLF_B, xflags + 8);          // [16][200] h1
    const ClX xC = cl_x(xbase + CLF_A + CLF_B, CLF_C, xflags + 16); // [16][300] output vector
    unsigned* tmo = a.cl_timeout;

    for (int i = tid; i < TR * LDX; i += NTHR) xb[i] = (bf16)0.f;
    for (int i = tid; i < 3 * TR * LDH; i += NTHR) h0b[i] = (bf16)0.f;
    for (int i = tid; i < G; i += NTHR) { bias[i] = a.bhh0[i]; bias[G + i] = a.bih1[i]; bias[2 * G + i] = a.bhh1[i]; }
    __syncthreads();
    for (int i = tid; i < TR * H; i += NTHR) {
        const int row = i / H, j = i - row * H;
        const float v = r0 + row < R ? a.hinit[(size_t)(r0 + row) * H + j] : 0.f;
        h0b[row * LDH + j] = (bf16)v; h1b[row * LDH + j] = (bf16)v;
        if (SAVE && rank == 0 && r0 + row < R) {
            a.h0b_all[(size_t)(r0 + row) * HP + j] = (bf16)v; a.h1b_all[(size_t)(r0 + row) * HP + j] = (bf16)v;
        }
    }
    for (int i = tid; i < TR * E; i += NTHR) {
        const int row = i / E, e = i - row * E;
        const bf16 v = (bf16)a.sos[e];
        xb[row * LDX + e] = v;
        if (SAVE && rank == 0 && r0 + row < R) a.xb_all[(size_t)(r0 + row) * XP + e] = v;
    }
    if (SAVE && rank == 0 && pl == 0 && gok) { a.h0b_all[gr * HP + H] = (bf16)1.f; a.h1b_all[gr * HP + H] = (bf16)1.f; }
    float h0f[2] = {0.f, 0.f}, h1f[2] = {0.f, 0.f};
    if (uok && gok) {
        h0f[0] = h1f[0] = a.hinit[gr * H + j0]; h0f[1] = h1f[1] = a.hinit[gr * H + j0 + 1];
    }
    const ClMat m_ih0{__builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.wg_ih0), 0, 39 * (XP / 32) * 1024, 0x00020000), XP / 32};
    const ClMat m_hh0{__builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.wg_hh0), 0, 39 * (HP / 32) * 1024, 0x00020000), HP / 32};
    const ClMat m_ih1{__builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.wg_ih1), 0, 39 * (HP / 32) * 1024, 0x00020000), HP / 32};
    const ClMat m_hh1{__builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.wg_hh1), 0, 39 * (HP / 32) * 1024, 0x00020000), HP / 32};
    const ClMat m_ho{__builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_ho), 0, 19 * (HP / 32) * 1024, 0x00020000), HP / 32};
    bf16x8 ring[D][KCH];
    auto request = [&](const ClMat& m, bool gates, int n) {      // the chunks of ONE product (its tile slots) into the ring
#pragma unroll
        for (int q = 0; q < D; ++q) { if (gates) cl_load<true, P>(ring[q], m, q, n, rank, wave, lane); else cl_load<false, P>(ring[q], m, q, n, rank, wave, lane); }
    };
    const float* zi0 = a.zi0 + (size_t)r0 * G;
    const float* zo = a.zo + (size_t)r0 * E;
    // Order of a step.  The hidden-state products do not depend on the exchange that is in flight when they run, so each of
    // them hides one:   ih0 x | gates 0, publish A | hh1 h1 (during A) | ih1 mid | gates 1, publish B | hh0 h0' for the NEXT
    // step (during B) | ho h1' | output, publish C | (wait C).  One product's chunks sit in the ring at a time, requested
    // right behind the flag store (overlapped products) or behind the loads of the exchange in front of the product.
    request(m_hh0, true, ng);
    __syncthreads();
    cl_gemm<HP / 32, MT, MT, D, 0, true, true, P, false>(h0b, LDH, m_hh0, ng, gb, LDC, ring, m_ih0, ng, true, rank, wave, lane, nullptr, 0, 0, H, rows_ok);
    for (int t = 0; t < T; ++t) {
        const unsigned ep = 3u * (unsigned)t + 1u;
        const bool last = t + 1 == T;
        // ---- layer 0 (its hidden product is in gb already)
        cl_gemm<XP / 32, MT, MT, D, 0, true, true, P, true>(xb, LDX, m_ih0, ng, ga, LDC, ring, m_hh1, ng, false, rank, wave, lane, zi0, G, H, H, rows_ok);
        unsigned short kpo = 0x0101;
        if (KEEP && uok) kpo = gok ? *reinterpret_cast<const unsigned short*>(a.keep + (size_t)t * RH + gr * H + j0) : (unsigned short)0;
        __syncthreads();
        if (uok) {
            bf16 pub[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = j0 + q, lc = (3 * uk) * 16 + uu + q;          // local column of gate g: lc + 16 g
                const float* pa = ga + grow * LDC; const float* pb = gb + grow * LDC;
                const float r = sigm(pa[lc] + pb[lc] + bias[j]);
                const float z = sigm(pa[lc + 16] + pb[lc + 16] + bias[H + j]);
                const float ghn = pb[lc + 32] + bias[2 * H + j];
                const float n = tanh_fast(pa[lc + 32] + r * ghn);
                const float hn = (1.0f - z) * n + z * h0f[q];
                h0f[q] = hn;
                float mid = hn;
                if (KEEP) mid = ((kpo >> (8 * q)) & 0xff) ? hn * a.keep_scale : 0.f;
                if (SAVE && gok) {
                    a.h0_all[(size_t)(t + 1) * RH + gr * H + j] = hn;
                    float* s = a.sav0 + ((size_t)t * R + gr) * 4 * H;
                    s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
                    a.h0b_all[((size_t)(t + 1) * R + gr) * HP + j] = (bf16)hn;
                    a.midb_all[((size_t)t * R + gr) * HP + j] = (bf16)mid;
                }
                pub[2 * q] = (bf16)hn; pub[2 * q + 1] = (bf16)mid;
            }
            cl_store8(xA, (grow * H + j0) * 4, pub[0], pub[1], pub[2], pub[3]);      // (row, unit pair) = {h0, dropout(h0)} x 2
        }
        if (SAVE && rank == 0 && pl == 0 && gok) {
            a.h0b_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
            a.midb_all[((size_t)t * R + gr) * HP + H] = (bf16)1.f;
            a.h1b_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
        }
        // all-gather: every rank's units of the new h0 and of mid (own ones included: same path)
        cl_signal(xA, rank, ep, tid);
        request(m_hh1, true, ng);        // layer 1's hidden product reads last step's h1: it runs while exchange A completes
        cl_gemm<HP / 32, MT, MT, D, 0, true, true, P, false>(h1b, LDH, m_hh1, ng, gb, LDC, ring, m_ih1, ng, false, rank, wave, lane, nullptr, 0, 0, H, rows_ok);
        cl_wait<P>(xA, ep, tmo, tid);
        {   // 16 bytes = 4 units x {h0, mid}.  All pieces are requested, THEN the weights of the next phase (they are not in
            // flight while the payload stores drain, and they travel while the exchange completes), then the pieces are used
            constexpr int NPC = (TR * H / 4 + NTHR - 1) / NTHR;
            bf16x8 x[NPC];
#pragma unroll
            for (int q = 0; q < NPC; ++q) { const int v = tid + q * NTHR; if (v < TR * H / 4) x[q] = cl_load16(xA, v * 16); }
#pragma unroll
            for (int q = 0; q < D; ++q) cl_load<true, P>(ring[q], m_ih1, q, ng, rank, wave, lane);
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int v = tid + q * NTHR;
                if (v < TR * H / 4) {
                    const int row = v / (H / 4), j = (v - row * (H / 4)) * 4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { h0b[row * LDH + j + u] = x[q][2 * u]; midb[row * LDH + j + u] = x[q][2 * u + 1]; }
                }
            }
        }
        __syncthreads();
        // ---- layer 1
        cl_gemm<HP / 32, MT, MT, D, 0, true, true, P, false>(midb, LDH, m_ih1, ng, ga, LDC, ring, m_hh1, ng, false, rank, wave, lane, nullptr, 0, 0, H, rows_ok);
        __syncthreads();
        if (uok) {
            bf16 hb2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = j0 + q, lc = (3 * uk) * 16 + uu + q;
                const float* pa = ga + grow * LDC; const float* pb = gb + grow * LDC;
                const float r = sigm(pa[lc] + bias[G + j] + pb[lc] + bias[2 * G + j]);
                const float z = sigm(pa[lc + 16] + bias[G + H + j] + pb[lc + 16] + bias[2 * G + H + j]);
                const float ghn = pb[lc + 32] + bias[2 * G + 2 * H + j];
                const float n = tanh_fast(pa[lc + 32] + bias[G + 2 * H + j] + r * ghn);
                const float hn = (1.0f - z) * n + z * h1f[q];
                h1f[q] = hn; hb2[q] = (bf16)hn;
                if (SAVE && gok) {
                    a.h1_all[(size_t)(t + 1) * RH + gr * H + j] = hn;
                    float* s = a.sav1 + ((size_t)t * R + gr) * 4 * H;
                    s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
                    a.h1b_all[((size_t)(t + 1) * R + gr) * HP + j] = (bf16)hn;
                }
            }
            cl_store4(xB, (grow * H + j0) * 2, hb2[0], hb2[1]);
        }
        cl_signal(xB, rank, ep + 1, tid);
        if (!last) {                     // layer 0's hidden product of the NEXT step reads the h0 of exchange A: it runs during B
            request(m_hh0, true, ng);
            cl_gemm<HP / 32, MT, MT, D, 0, true, true, P, false>(h0b, LDH, m_hh0, ng, gb, LDC, ring, m_ih0, ng, false, rank, wave, lane, nullptr, 0, 0, H, rows_ok);
        }
        cl_wait<P>(xB, ep + 1, tmo, tid);
        {   // 16 bytes = 8 units (400 pieces: one per thread)
            bf16x8 x = {};
            if (tid < TR * H / 8) x = cl_load16(xB, tid * 16);
#pragma unroll
            for (int q = 0; q < MTO; ++q) cl_load<false, P>(ring[q], m_ho, q, noe, rank, wave, lane);
            if (tid < TR * H / 8) {
                const int row = tid / (H / 8), j = (tid - row * (H / 8)) * 8;
                *reinterpret_cast<bf16x8*>(h1b + row * LDH + j) = x;
            }
        }
        __syncthreads();
        // ---- output projection (own embedding tiles), fed back as the next input
        cl_gemm<HP / 32, MT, MT, D, 0, false, true, P, true>(h1b, LDH, m_ho, noe, ga, LDC, ring, m_ih0, ng, false, rank, wave, lane, zo, E, E, 0, rows_ok);
        __syncthreads();
        // own columns: local pair lp -> tile lp / 8, columns 2 (lp % 8), +1 ; up to NOEMAX * 8 pairs per row
        for (int gi = tid; gi < TR * NOEMAX * 8; gi += NTHR) {
            const int row = gi / (NOEMAX * 8), lp = gi - row * (NOEMAX * 8);
            const int lt = lp >> 3, e = (rank + P * lt) * 16 + 2 * (lp & 7);
            if (lt < noe && e < E) {
                const float v0 = ga[row * LDC + lt * 16 + 2 * (lp & 7)], v1 = ga[row * LDC + lt * 16 + 2 * (lp & 7) + 1];
                if (r0 + row < R) {
                    float* sp = a.sentence + ((size_t)(r0 + row) * T + t) * E + e;
                    sp[0] = v0; sp[1] = v1;
                    if (SAVE && !last) {
                        bf16* xp = a.xb_all + ((size_t)(t + 1) * R + r0 + row) * XP + e;
                        xp[0] = (bf16)v0; xp[1] = (bf16)v1;
                    }
                }
                if (!last) cl_store4(xC, (row * E + e) * 2, (bf16)v0, (bf16)v1);
            }
        }
        if (!last) {
            cl_signal(xC, rank, ep + 2, tid);
            cl_wait<P>(xC, ep + 2, tmo, tid);
            // 8 bytes = 4 columns (E * 2 bytes per row is not a multiple of 16)
            constexpr int NPC = (TR * E / 4 + NTHR - 1) / NTHR;
            u32x2 x[NPC];
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int v = tid + q * NTHR;
                if (v < TR * E / 4) { const int row = v / (E / 4), e = (v - row * (E / 4)) * 4; x[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xC.rs, (row * E + e) * 2, 0, 16)); }
            }
#pragma unroll
            for (int q = 0; q < D; ++q) cl_load<true, P>(ring[q], m_ih0, q, ng, rank, wave, lane);
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int v = tid + q * NTHR;
                if (v < TR * E / 4) { const int row = v / (E / 4), e = (v - row * (E / 4)) * 4; *reinterpret_cast<u32x2*>(xb + row * LDX + e) = x[q]; }
            }
        }
        __syncthreads();
    }
}

// ================================================================== forward, COMPOSED cluster form (8 ranks, weights resident)
// The decoder feeds its own output back: x[t+1] = W_ho h1[t] + zo, and the only consumer of x[t+1] inside the recurrence is
// layer 0's input projection W_ih0x x[t+1].  With W_comb = W_ih0x W_ho ([600][200], recomputed from the fp32 parameters by
// coco_comb_kernel whenever they change) that projection is W_comb h1[t] + (W_ih0x zo + zi0): it reads the h1 every rank holds
// after exchange B, so the third all-gather of a step (the output vector) disappears from the chain, and the output projection
// itself -- still needed: the sentence is the result, and its bf16 copy the operand of the weight gradient -- moves into the
// window of exchange A of the NEXT step.  K of the composed product is 224 instead of 320, and with 8 ranks a wave owns ONE
// tile of each of the five matrices: 5 x 7 k-steps x 4 VGPRs = 140 registers hold every weight the workgroup ever needs, so
// nothing streams inside the loop.  Step: gates 0, publish A | hh1 h1 and ho h1 (step t-1's output) during A | ih1 mid |
// gates 1, publish B | hh0 h0' and the output pass of step t-1 during B | comb h1'.
template <bool KEEP, bool SAVE>
__global__ __launch_bounds__(NTHR) void coco_dec_fwd_c8_kernel(const CocoDecFwdArgs a) {
    constexpr int P = 8, NUBMAX = 2, NOEMAX = 3, KS = HP / 32;
    constexpr int LDC = 3 * NUBMAX * 16 + 4, LDOE = NOEMAX * 16 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ga = reinterpret_cast<float*>(smem);                     // [16][LDC] input-side gate pre-activations (own units)
    float* gb = ga + TR * LDC;                                      // [16][LDC] hidden-side
    float* go = gb + TR * LDC;                                      // [16][LDOE] own output columns
    float* bias = go + TR * LDOE;                                   // b_hh0 | b_ih1 | b_hh1, [3][G]
    bf16* h0b = reinterpret_cast<bf16*>(bias + 3 * G);              // [16][LDH]
    bf16* midb = h0b + TR * LDH;
    bf16* h1b = midb + TR * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nblk = (a.R + TR - 1) / TR, nblk_pad = (nblk + 7) / 8 * 8;
    const int rank = blockIdx.x / nblk_pad, blk = blockIdx.x - rank * nblk_pad;
    if (blk >= nblk) return;
    const int r0 = blk * TR, R = a.R, T = a.T;
    const size_t RH = (size_t)R * H;
    const int rows_ok = min(TR, R - r0);
    const int nub = (13 - rank + P - 1) / P, noe = (19 - rank + P - 1) / P;
    const int ng = 3 * nub;
    const int grow = tid >> 5, pl = tid & 31;
    const bool gok = r0 + grow < R;
    const size_t gr = gok ? r0 + grow : 0;
    // ONE own unit per thread (a rank has at most 32): local column lc (+ 16 per gate) of the gate planes
    const int uk = pl >> 4, ju = (rank + P * uk) * 16 + (pl & 15), lc = 3 * uk * 16 + (pl & 15);
    const bool uok = uk < nub && ju < H;
    char* xbase = reinterpret_cast<char*>(a.cl_xchg) + (size_t)blk * CLF_BYTES;
    unsigned* xflags = reinterpret_cast<unsigned*>(xbase + CLF_A + CLF_B + CLF_C);
    const ClX xA = cl_x(xbase, CLF_A, xflags);                      // [16][200] {h0, dropout(h0)}
    const ClX xB = cl_x(xbase + CLF_A, CLF_B, xflags + 8);          // [16][200] h1
    unsigned* tmo = a.cl_timeout;

    for (int i = tid; i < 3 * TR * LDH; i += NTHR) h0b[i] = (bf16)0.f;
    for (int i = tid; i < G; i += NTHR) { bias[i] = a.bhh0[i]; bias[G + i] = a.bih1[i]; bias[2 * G + i] = a.bhh1[i]; }
    __syncthreads();
    for (int i = tid; i < TR * H; i += NTHR) {
        const int row = i / H, j = i - row * H;
        const float v = r0 + row < R ? a.hinit[(size_t)(r0 + row) * H + j] : 0.f;
        h0b[row * LDH + j] = (bf16)v; h1b[row * LDH + j] = (bf16)v;
        if (SAVE && rank == 0 && r0 + row < R) {
            a.h0b_all[(size_t)(r0 + row) * HP + j] = (bf16)v; a.h1b_all[(size_t)(r0 + row) * HP + j] = (bf16)v;
            a.h1_all[(size_t)(r0 + row) * H + j] = v;       // (slice 0 of h0_all IS hinit; this form's caller leaves h1's to us)
        }
    }
    if (SAVE && rank == 0)
        for (int i = tid; i < TR * E; i += NTHR) {
            const int row = i / E, e = i - row * E;
            if (r0 + row < R) a.xb_all[(size_t)(r0 + row) * XP + e] = (bf16)a.sos[e];
        }
    if (SAVE && rank == 0 && pl == 0 && gok) { a.h0b_all[gr * HP + H] = (bf16)1.f; a.h1b_all[gr * HP + H] = (bf16)1.f; }
    float h0f = 0.f, h1f = 0.f;
    if (uok && gok) h0f = h1f = a.hinit[gr * H + ju];
    // ---- resident weights: this wave's tile of the four gate matrices (tile list index = wave) and of the output projection
    // (list index 7 - wave: the waves the gate products leave idle)
    const bool gact = wave < ng;
    const int lo = NW - 1 - wave;
    const bool owave = lo < noe;
    bf16x8 wc[KS], w_hh0[KS], w_ih1[KS], w_hh1[KS], w_o[KS];
    {
        auto rs = [](const bf16* p, int tiles) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p), 0, tiles * KS * 1024, 0x00020000); };
        const __amdgpu_buffer_rsrc_t r_c = rs(a.wg_comb, 39), r_0 = rs(a.wg_hh0, 39), r_1 = rs(a.wg_ih1, 39), r_2 = rs(a.wg_hh1, 39), r_o = rs(a.w_ho, 19);
        const int gt = gact ? ((wave % 3) * 13 + rank + P * (wave / 3)) * KS * 1024 : 0;
        const int ot = owave ? (rank + P * lo) * KS * 1024 : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            wc[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_c, lane * 16, gt + s * 1024, 0));
            w_hh0[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_0, lane * 16, gt + s * 1024, 0));
            w_ih1[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_1, lane * 16, gt + s * 1024, 0));
            w_hh1[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_2, lane * 16, gt + s * 1024, 0));
            w_o[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_o, lane * 16, ot + s * 1024, 0));
        }
    }
    // time-invariant addends of the two products with one, in the accumulator layout (rows 4 fq + j, column fr of the tile)
    f32x4 zc = {0.f, 0.f, 0.f, 0.f}, zof = {0.f, 0.f, 0.f, 0.f};
    {
        const int within = (rank + P * (wave / 3)) * 16 + fr;
        if (gact && within < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) zc[j] = a.zi0p[(size_t)(r0 + min(fq * 4 + j, rows_ok - 1)) * G + (wave % 3) * H + within];
        }
        const int e = (rank + P * lo) * 16 + fr;
        if (owave && e < E) {
#pragma unroll
            for (int j = 0; j < 4; ++j) zof[j] = a.zo[(size_t)(r0 + min(fq * 4 + j, rows_ok - 1)) * E + e];
        }
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto rgemm = [&](const bf16* A, const bf16x8 (&w)[KS], f32x4 acc, float* out, int ldo, int li) {
        bf16x8 af[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const bf16x8*>(A + fr * LDH + s * 32 + fq * 8);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], w[s], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(fq * 4 + j) * ldo + li * 16 + fr] = acc[j];
    };
    // own output columns of step ts: go -> the sentence and (as the bf16 operand of the weight gradient) slice ts + 1 of xb_all.
    // With a.mse_target the reconstruction loss rides along (coco/train.py:150-158: MSE against the caption's word vectors): the
    // squared error is accumulated per thread, its gradient written in the two forms the BPTT kernel reads -- one pass over the
    // sentence less between the two recurrences.
    const int orow = tid / (NOEMAX * 8), olp = tid - orow * (NOEMAX * 8);
    const int oe = (rank + P * (olp >> 3)) * 16 + 2 * (olp & 7);
    const bool oact = tid < TR * NOEMAX * 8 && (olp >> 3) < noe && oe < E && r0 + orow < R;
    const bool mse = a.mse_target != nullptr;
    const int ogrp = oact && mse ? (r0 + orow) / a.mse_B : 0;
    const float* tgt = mse ? a.mse_target + ((size_t)(r0 + orow - ogrp * a.mse_B) * T) * E + oe : nullptr;
    const float c2 = mse ? 2.f * a.mse_coef[ogrp] : 0.f;
    float sq = 0.f, tg[2] = {0.f, 0.f};
    auto get_target = [&](int ts) { if (mse && oact) { tg[0] = tgt[(size_t)ts * E]; tg[1] = tgt[(size_t)ts * E + 1]; } };
    auto put_output = [&](int ts) {
        if (oact) {
            const float v0 = go[orow * LDOE + (olp >> 3) * 16 + 2 * (olp & 7)], v1 = go[orow * LDOE + (olp >> 3) * 16 + 2 * (olp & 7) + 1];
            float* sp = a.sentence + ((size_t)(r0 + orow) * T + ts) * E + oe;
            sp[0] = v0; sp[1] = v1;
            if (SAVE && ts + 1 < T) {
                bf16* xp = a.xb_all + ((size_t)(ts + 1) * R + r0 + orow) * XP + oe;
                xp[0] = (bf16)v0; xp[1] = (bf16)v1;
            }
            if (mse) {
                const float d0 = v0 - tg[0], d1 = v1 - tg[1];
                sq += d0 * d0 + d1 * d1;
                if (a.mse_dw) {
                    float* dp = a.mse_dw + ((size_t)(r0 + orow) * T + ts) * E + oe;
                    dp[0] = c2 * d0; dp[1] = c2 * d1;
                    bf16* hp = a.mse_dw16 + ((size_t)(r0 + orow) * T + ts) * XP + oe;
                    hp[0] = (bf16)(c2 * d0); hp[1] = (bf16)(c2 * d1);
                }
            }
        }
    };
    if (mse && a.mse_dw && rank == 0)       // pad columns of the bf16 gradient rows (K padding of dw W_ho): zero
        for (int i = tid; i < rows_ok * T * (XP - E); i += NTHR) {
            const int rt = i / (XP - E), c = i - rt * (XP - E);
            a.mse_dw16[((size_t)r0 * T + rt) * XP + E + c] = (bf16)0.f;
        }
    // step 0: the input is '<s>' for every row: ga = zi0 + W_ih0x sos (a.sosv, made with W_comb); gb = W_hh0 h(init)
    for (int i = tid; i < TR * 3 * NUBMAX * 16; i += NTHR) {
        const int row = i / (3 * NUBMAX * 16), c = i - row * (3 * NUBMAX * 16);
        const int li = c >> 4, within = (rank + P * (li / 3)) * 16 + (c & 15);
        float v = 0.f;
        if (li < ng && within < H) {
            const int col = (li % 3) * H + within;
            v = a.zi0[(size_t)(r0 + min(row, rows_ok - 1)) * G + col] + a.sosv[col];
        }
        ga[row * LDC + c] = v;
    }
    __syncthreads();
    if (gact) rgemm(h0b, w_hh0, zero4, gb, LDC, wave);
    __syncthreads();
    // Loads of the loop are requested right BEHIND a flag store and used a phase later: cl_signal drains the wave's memory
    // queue (vmcnt is in order), so a load in flight across it delays the flag by its own latency, and a load used soon
    // after a burst of saves waits for their acknowledgements.  The dropout keep flag of the own unit, read where it is used,
    // sat with its L2 / HBM round trip in front of layer 0's gate math on every step (timed with s_memtime: 1,100-2,600 of a
    // step's 12,000 clocks); now it is requested behind the flag of exchange B of the step before.
    unsigned kp_next = 1;
    if (KEEP && uok) kp_next = gok ? a.keep[gr * H + ju] : 0;
    for (int t = 0; t < T; ++t) {
        const unsigned ep = (unsigned)t + 1u;
        const bool last = t + 1 == T;
        // ---- layer 0
        {
            const unsigned kp = kp_next;
            float sr = 0.f, sz = 0.f, sn = 0.f, sg = 0.f, sm = 0.f;
            if (uok) {
                const float* pa = ga + grow * LDC; const float* pb = gb + grow * LDC;
                sr = sigm(pa[lc] + pb[lc] + bias[ju]);
                sz = sigm(pa[lc + 16] + pb[lc + 16] + bias[H + ju]);
                sg = pb[lc + 32] + bias[2 * H + ju];
                sn = tanh_fast(pa[lc + 32] + sr * sg);
                h0f = (1.0f - sz) * sn + sz * h0f;
                sm = h0f;
                if (KEEP) sm = kp ? h0f * a.keep_scale : 0.f;
                cl_store4(xA, (grow * H + ju) * 4, (bf16)h0f, (bf16)sm);       // (row, unit) = {h0, dropout(h0)}
            }
            cl_signal(xA, rank, ep, tid);
            if (t > 0) get_target(t - 1);    // (for the output pass behind exchange A)
            if (SAVE && uok && gok) {        // (behind the flag: the exchange does not wait for these)
                a.h0_all[(size_t)(t + 1) * RH + gr * H + ju] = h0f;
                float* s = a.sav0 + ((size_t)t * R + gr) * 4 * H;
                s[ju] = sr; s[H + ju] = sz; s[2 * H + ju] = sn; s[3 * H + ju] = sg;
                a.h0b_all[((size_t)(t + 1) * R + gr) * HP + ju] = (bf16)h0f;
                a.midb_all[((size_t)t * R + gr) * HP + ju] = (bf16)sm;
            }
            if (SAVE && rank == 0 && pl == 0 && gok) {
                a.h0b_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
                a.midb_all[((size_t)t * R + gr) * HP + H] = (bf16)1.f;
                a.h1b_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
            }
        }
        // during exchange A: layer 1's hidden product and the output projection of the previous step (both read last step's h1)
        if (gact) rgemm(h1b, w_hh1, zero4, gb, LDC, wave);
        if (t > 0 && owave) rgemm(h1b, w_o, zof, go, LDOE, lo);
        cl_wait<P>(xA, ep, tmo, tid);
        {   // 16 bytes = 4 units x {h0, mid}
            constexpr int NPC = (TR * H / 4 + NTHR - 1) / NTHR;
            bf16x8 x[NPC];
#pragma unroll
            for (int q = 0; q < NPC; ++q) { const int v = tid + q * NTHR; if (v < TR * H / 4) x[q] = cl_load16(xA, v * 16); }
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int v = tid + q * NTHR;
                if (v < TR * H / 4) {
                    const int row = v / (H / 4), j = (v - row * (H / 4)) * 4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { h0b[row * LDH + j + u] = x[q][2 * u]; midb[row * LDH + j + u] = x[q][2 * u + 1]; }
                }
            }
        }
        __syncthreads();
        // ---- layer 1
        if (gact) rgemm(midb, w_ih1, zero4, ga, LDC, wave);
        __syncthreads();
        {
            float sr = 0.f, sz = 0.f, sn = 0.f, sg = 0.f;
            if (uok) {
                const float* pa = ga + grow * LDC; const float* pb = gb + grow * LDC;
                sr = sigm(pa[lc] + bias[G + ju] + pb[lc] + bias[2 * G + ju]);
                sz = sigm(pa[lc + 16] + bias[G + H + ju] + pb[lc + 16] + bias[2 * G + H + ju]);
                sg = pb[lc + 32] + bias[2 * G + 2 * H + ju];
                sn = tanh_fast(pa[lc + 32] + bias[G + 2 * H + ju] + sr * sg);
                h1f = (1.0f - sz) * sn + sz * h1f;
                const bf16 hb = (bf16)h1f;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, hb), xB.rs, (grow * H + ju) * 2, 0, 16);
            }
            cl_signal(xB, rank, ep, tid);
            if (KEEP && uok && !last) kp_next = gok ? a.keep[(size_t)(t + 1) * RH + gr * H + ju] : 0;
            if (SAVE && uok && gok) {
                a.h1_all[(size_t)(t + 1) * RH + gr * H + ju] = h1f;
                float* s = a.sav1 + ((size_t)t * R + gr) * 4 * H;
                s[ju] = sr; s[H + ju] = sz; s[2 * H + ju] = sn; s[3 * H + ju] = sg;
                a.h1b_all[((size_t)(t + 1) * R + gr) * HP + ju] = (bf16)h1f;
            }
        }
        // during exchange B: layer 0's hidden product of the NEXT step (reads the h0 of exchange A), and the output pass of the
        // step before (its projection ran during exchange A: barriers in between, and this window has the slack)
        if (!last && gact) rgemm(h0b, w_hh0, zero4, gb, LDC, wave);
        if (t > 0) put_output(t - 1);
        cl_wait<P>(xB, ep, tmo, tid);
        if (tid < TR * H / 8) {     // 16 bytes = 8 units (400 pieces: one per thread)
            const bf16x8 x = cl_load16(xB, tid * 16);
            const int row = tid / (H / 8), j = (tid - row * (H / 8)) * 8;
            *reinterpret_cast<bf16x8*>(h1b + row * LDH + j) = x;
        }
        __syncthreads();
        // ---- layer 0's input projection of the next step, straight from h1
        if (!last && gact) rgemm(h1b, wc, zc, ga, LDC, wave);
        __syncthreads();
    }
    get_target(T - 1);
    if (owave) rgemm(h1b, w_o, zof, go, LDOE, lo);
    __syncthreads();
    put_output(T - 1);
    if (mse) {      // squared error: rows of the block meet in LDS, one atomic per row into the row's pass
        float* red = ga;
        __syncthreads();
        if (tid < TR) red[tid] = 0.f;
        __syncthreads();
        if (oact) atomicAdd(red + orow, sq);
        __syncthreads();
        if (tid < rows_ok) atomicAdd(a.mse_loss + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + 4 + (r0 + tid) / a.mse_B, red[tid]);
    }
}

// W_comb = W_ih0[:, :300] W_ho[:, :200] in fp32, written as the per-gate forward packs [3][208][224] and the transposed pack
// [208][608] (fragment-major, pads zero); sosv = W_ih0[:, :300] sos; and the z-side of the same composition, fp32:
// wz = W_ih0[:, 300:] + W_ih0[:, :300] W_ho[:, 200:] ([600][D]), bz = b_ih0 + W_ih0[:, :300] b_ho.  4 gate rows per workgroup,
// thread = column of [W_ho | its z part].
constexpr int COMB_TPB = 320;
__global__ __launch_bounds__(COMB_TPB) void coco_comb_kernel(const float* __restrict__ wih0, const float* __restrict__ bih0, const float* __restrict__ who,
                                                             const float* __restrict__ bho, int D, const float* __restrict__ sos, bf16* comb, bf16* combT,
                                                             float* sosv, float* wz, float* bz) {
    __shared__ float wr[4][E];
    const int in0 = E + D, ino = H + D;
    const int c0 = blockIdx.x * 4, k = threadIdx.x;
    for (int i = k; i < 4 * E; i += COMB_TPB) {
        const int r = i / E, e = i - r * E;
        wr[r][e] = c0 + r < G ? wih0[(size_t)(c0 + r) * in0 + e] : 0.f;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < ino) {
#pragma unroll 4
        for (int e = 0; e < E; ++e) {
            const float v = who[(size_t)e * ino + k];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += wr[r][e] * v;
        }
    }
    auto frag = [](int n, int kk, int kpad) {
        const int kv = kk >> 3;
        return ((size_t)((n >> 4) * (kpad >> 5) + (kv >> 2)) * 64 + (kv & 3) * 16 + (n & 15)) * 8 + (kk & 7);
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + r;
        if (k >= H) {                        // z side
            if (c < G && k < ino) wz[(size_t)c * D + (k - H)] = wih0[(size_t)c * in0 + E + (k - H)] + acc[r];
            continue;
        }
        if (c < G) {
            const int g = c / H, n = c - g * H;
            comb[(size_t)g * 208 * HP + frag(n, k, HP)] = (bf16)acc[r];
        } else if (c < GP) {                 // rows 200..207 of the three gates: zero
            const int n = H + (c - G);
            for (int g = 0; g < 3; ++g) comb[(size_t)g * 208 * HP + frag(n, k, HP)] = (bf16)0.f;
        }
        if (c < GP) combT[frag(k, c, GP)] = (bf16)(c < G ? acc[r] : 0.f);
    }
    // pad columns / rows of the two packs (k = 200..223 of comb, unit rows 200..207 of combT): threads 0..23 / 0..7
    if (k < HP - H)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + r;
            if (c < G) { const int g = c / H, n = c - g * H; comb[(size_t)g * 208 * HP + frag(n, H + k, HP)] = (bf16)0.f; }
            else if (c < GP) for (int g = 0; g < 3; ++g) comb[(size_t)g * 208 * HP + frag(H + (c - G), H + k, HP)] = (bf16)0.f;
            if (k < 208 - H && c < GP) combT[frag(H + k, c, GP)] = (bf16)0.f;
        }
    const int w = k >> 6, lane = k & 63, c = c0 + w;
    if (w < 4 && c < G) {
        float t = 0.f, u = 0.f;
        for (int e = lane; e < E; e += 64) { t += wr[w][e] * sos[e]; u += wr[w][e] * bho[e]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o); u += __shfl_xor(u, o); }
        if (lane == 0) { sosv[c] = t; bz[c] = bih0[c] + u; }
    }
}

// ================================================================== backward (BPTT)
// Thread (row, c0) keeps the time sum of the output gradient of its 10 columns in registers; the time sum of the layer-0
// input-projection gradient (what the z-columns and the bias see) is taken from the saved operand afterwards.
template <bool KEEP>
__global__ __launch_bounds__(NTHR) void coco_dec_bwd_kernel(const CocoDecBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dh0f = reinterpret_cast<float*>(smem);                   // [16][H] gradient carried into h0[t]
    float* dh1f = dh0f + TR * H;
    float* fb = dh1f + TR * H;                                      // [16][LDO] feedback gradient into the previous output
    float* o1 = fb + TR * LDO;                                      // [16][LDT] GEMM results (200 hidden units)
    float* o2 = o1 + TR * LDT;
    bf16* dob = reinterpret_cast<bf16*>(o2 + TR * LDT);             // [16][LDX] total gradient wrt this step's output
    bf16* dgi = dob + TR * LDX;                                     // [16][LDGK]
    bf16* dgh = dgi + TR * LDGK;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR
    const int r0 = blockIdx.x * TR, R = a.R, T = a.T;
    const size_t RH = (size_t)R * H;
    const int grow = tid >> 5, c0 = tid & 31;
    const bool gok = r0 + grow < R;
    const size_t gr = gok ? r0 + grow : 0;
    for (int i = tid; i < 2 * TR * H; i += NTHR) dh0f[i] = 0.f;
    for (int i = tid; i < TR * LDO; i += NTHR) fb[i] = 0.f;
    for (int i = tid; i < TR * LDX; i += NTHR) dob[i] = (bf16)0.f;
    for (int i = tid; i < 2 * TR * LDGK; i += NTHR) dgi[i] = (bf16)0.f;
    float ws_sum[NQE];
#pragma unroll
    for (int i = 0; i < NQE; ++i) ws_sum[i] = 0.f;
    // chunk schedule of a step (ring depth 2): hoT 2, ih1T 4, hh1T 4, hh0T 4, ih0T 6 = 20 chunks (every GEMM starts in slot 0)
    constexpr int D = 2;
    const WMat<XP / 32, HP / 16 - 1> d_hoT(a.w_hoT);
    const WMat<GP / 32, HP / 16 - 1> d_ih1T(a.w_ih1T), d_hh1T(a.w_hh1T), d_hh0T(a.w_hh0T);
    const WMat<GP / 32, EP / 16> d_ih0T(a.w_ih0T);
    bf16x8 ring[D][KCH];
#pragma unroll
    for (int q = 0; q < D; ++q) load_chunk(ring[q], d_hoT, q, wave, lane);
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        // ---- total gradient wrt the output of step t: loss term + what step t+1 sent back through its input
        {
            float dwv[NQE];
#pragma unroll
            for (int q = 0; q < NQE; ++q) {
                const int e = c0 + 32 * q;
                dwv[q] = gok && e < E ? a.dw[(gr * T + t) * E + e] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < NQE; ++q) {
                const int e = c0 + 32 * q;
                if (e < E) {
                    const float v = dwv[q] + fb[grow * LDO + e];
                    ws_sum[q] += v;
                    dob[grow * LDX + e] = (bf16)v;
                    if (gok) a.dout_b[((size_t)t * R + gr) * EP + e] = (bf16)v;
                }
            }
        }
        __syncthreads();
        // ---- dh1 += dOut * Who[:, :200]
        stream_gemm<2, D, 0>(dob, LDX, d_hoT, o1, LDT, ring, d_ih1T, true, wave, lane);
        // ---- layer-1 gates backward
        {
            float sr[NQH], sz[NQH], sn[NQH], sg[NQH], hp[NQH];
            const float* s = a.sav1 + ((size_t)t * R + gr) * 4 * H;
            const float* hpp = a.h1_all + (size_t)t * RH + gr * H;
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const int j = min(c0 + 32 * q, H - 1);
                sr[q] = s[j]; sz[q] = s[H + j]; sn[q] = s[2 * H + j]; sg[q] = s[3 * H + j]; hp[q] = hpp[j];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const int j = c0 + 32 * q;
                if (j < H) {
                    const int i = grow * H + j;
                    float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
                    if (gok) {
                        const float r = sr[q], z = sz[q], n = sn[q], ghn = sg[q];
                        const float d = dh1f[i] + o1[grow * LDT + j];
                        dn = d * (1.0f - z) * (1.0f - n * n);
                        dz = d * (hp[q] - n) * z * (1.0f - z);
                        dr = dn * ghn * r * (1.0f - r);
                        dnr = dn * r;
                        dd = d * z;
                    }
                    dh1f[i] = dd;
                    dgi[grow * LDGK + j] = (bf16)dr; dgi[grow * LDGK + H + j] = (bf16)dz; dgi[grow * LDGK + 2 * H + j] = (bf16)dn;
                    dgh[grow * LDGK + j] = (bf16)dr; dgh[grow * LDGK + H + j] = (bf16)dz; dgh[grow * LDGK + 2 * H + j] = (bf16)dnr;
                    if (gok) {
                        bf16* gi = a.dgi1_b + ((size_t)t * R + gr) * GP;
                        bf16* gh = a.dgh1_b + ((size_t)t * R + gr) * GP;
                        gi[j] = (bf16)dr; gi[H + j] = (bf16)dz; gi[2 * H + j] = (bf16)dn;
                        gh[j] = (bf16)dr; gh[H + j] = (bf16)dz; gh[2 * H + j] = (bf16)dnr;
                    }
                }
            }
        }
        __syncthreads();
        // ---- dmid = dgi1 * Wih1 ; dh1[t-1] += dgh1 * Whh1
        stream_gemm<2, D, 0>(dgi, LDGK, d_ih1T, o1, LDT, ring, d_hh1T, true, wave, lane);
        stream_gemm<2, D, 0>(dgh, LDGK, d_hh1T, o2, LDT, ring, d_hh0T, true, wave, lane);
        // ---- layer-0 gates backward (its output reached layer 1 through the dropout)
        {
            float sr[NQH], sz[NQH], sn[NQH], sg[NQH], hp[NQH];
            uint8_t kp[NQH];
            const float* s = a.sav0 + ((size_t)t * R + gr) * 4 * H;
            const float* hpp = a.h0_all + (size_t)t * RH + gr * H;
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const int j = min(c0 + 32 * q, H - 1);
                sr[q] = s[j]; sz[q] = s[H + j]; sn[q] = s[2 * H + j]; sg[q] = s[3 * H + j]; hp[q] = hpp[j];
                kp[q] = 1;
                if (KEEP) kp[q] = a.keep[(size_t)t * RH + gr * H + j];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NQH; ++q) {
                const int j = c0 + 32 * q;
                if (j < H) {
                    const int i = grow * H + j;
                    dh1f[i] += o2[grow * LDT + j];
                    float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
                    if (gok) {
                        const float r = sr[q], z = sz[q], n = sn[q], ghn = sg[q];
                        float dm = o1[grow * LDT + j];
                        if (KEEP) dm = kp[q] ? dm * a.keep_scale : 0.f;
                        const float d = dh0f[i] + dm;
                        dn = d * (1.0f - z) * (1.0f - n * n);
                        dz = d * (hp[q] - n) * z * (1.0f - z);
                        dr = dn * ghn * r * (1.0f - r);
                        dnr = dn * r;
                        dd = d * z;
                    }
                    dh0f[i] = dd;
                    dgi[grow * LDGK + j] = (bf16)dr; dgi[grow * LDGK + H + j] = (bf16)dz; dgi[grow * LDGK + 2 * H + j] = (bf16)dn;
                    dgh[grow * LDGK + j] = (bf16)dr; dgh[grow * LDGK + H + j] = (bf16)dz; dgh[grow * LDGK + 2 * H + j] = (bf16)dnr;
                    if (gok) {
                        bf16* gi = a.dgi0_b + ((size_t)t * R + gr) * GP;
                        bf16* gh = a.dgh0_b + ((size_t)t * R + gr) * GP;
                        gi[j] = (bf16)dr; gi[H + j] = (bf16)dz; gi[2 * H + j] = (bf16)dn;
                        gh[j] = (bf16)dr; gh[H + j] = (bf16)dz; gh[2 * H + j] = (bf16)dnr;
                    }
                }
            }
        }
        __syncthreads();
        // ---- dh0[t-1] += dgh0 * Whh0 ; feedback into the previous output = dgi0 * Wih0[:, :300]
        const bool first = t == 0;
        stream_gemm<2, D, 0>(dgh, LDGK, d_hh0T, o2, LDT, ring, d_ih0T, true, wave, lane);
        stream_gemm<3, D, 0>(dgi, LDGK, d_ih0T, fb, LDO, ring, d_hoT, !first, wave, lane);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = c0 + 32 * q;
            if (j < H) dh0f[grow * H + j] += o2[grow * LDT + j];
        }
        // (the next step's first phase touches fb / dob only; dh0f is next read after two more barriers)
    }
    __syncthreads();
    // ---- what is left: gradient of the shared initial state, time sum for the z-term of the output projection
    for (int i = tid; i < TR * H; i += NTHR) {
        const int row = i / H;
        if (r0 + row < R) a.dhinit[(size_t)(r0 + row) * H + (i - row * H)] = dh0f[i] + dh1f[i];
    }
#pragma unroll
    for (int q = 0; q < NQE; ++q) {
        const int e = c0 + 32 * q;
        if (gok && e < E) a.dwsum[gr * E + e] = ws_sum[q];
    }
}

// ================================================================== backward (BPTT), CLUSTER form
// The same decomposition as coco_dec_fwd_cl_kernel: rank r of the P workgroups of a row block owns the hidden-unit tiles
// r, r + P, ... of the four unit-indexed products (dOut W_ho, dgi1 W_ih1, dgh1 W_hh1, dgh0 W_hh0) and the embedding tiles
// r, r + P, ... of the feedback product dgi0 W_ih0.  A rank has only NUBMAX <= 4 tiles per product, so the reduction
// dimension of every tile is split over KH = 8 / NUBMAX waves whose partial tiles meet in LDS planes: all 8 waves stream,
// one chunk per wave and product.  Three all-gathers per step (the gate gradients of layer 1, of layer 0, and the total
// output gradient of the next step), tagged granules as in the forward kernel.
template <bool KEEP, int P>
__global__ __launch_bounds__(NTHR) void coco_dec_bwd_cl_kernel(const CocoDecBwdArgs a) {
    constexpr int NUBMAX = (13 + P - 1) / P, NOEMAX = (19 + P - 1) / P;
    constexpr int KH = NW / NUBMAX;                                   // K planes of the unit products
    constexpr int MTE = (NOEMAX * KH + NW - 1) / NW;                  // wave slots of the feedback product
    constexpr int NPOS = 4 + MTE;                                     // chunks per step and wave
    constexpr int D = NPOS % 2 == 0 ? 2 : 3;                          // (3 chunks ahead spill: 256 VGPRs)
    static_assert(NPOS % D == 0, "static ring slots");
    constexpr int KSG = GP / 32, KSX = XP / 32;                       // 19, 10 k-steps
    constexpr int KPG = (KSG + KH - 1) / KH, KPX = (KSX + KH - 1) / KH;   // k-steps per plane
    static_assert(KPG <= KCH && KPX <= KCH, "one chunk per plane");
    constexpr int LDU = NUBMAX * 16 + 4, LDE = NOEMAX * 16 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* o1 = reinterpret_cast<float*>(smem);                     // [KH][16][LDU]
    float* o2 = o1 + KH * TR * LDU;                                 // [KH][16][LDU]
    float* o3 = o2 + KH * TR * LDU;                                 // [KH][16][LDU]
    float* fbp = o3 + KH * TR * LDU;                                // [KH][16][LDE]
    bf16* dob = reinterpret_cast<bf16*>(fbp + KH * TR * LDE);       // [16][LDX]
    bf16* dgi = dob + TR * LDX;                                     // [16][LDGK]
    bf16* dgh = dgi + TR * LDGK;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nblk = (a.R + TR - 1) / TR, nblk_pad = (nblk + 7) / 8 * 8;
    const int rank = blockIdx.x / nblk_pad, blk = blockIdx.x - rank * nblk_pad;
    if (blk >= nblk) return;
    const int r0 = blk * TR, R = a.R, T = a.T;
    const size_t RH = (size_t)R * H;
    const int nub = (13 - rank + P - 1) / P, noe = (19 - rank + P - 1) / P;
    const int grow = tid >> 5, pl = tid & 31;
    const bool gok = r0 + grow < R;
    const size_t gr = gok ? r0 + grow : 0;
    const int uk = (2 * pl) >> 4, uu = (2 * pl) & 15;
    const int j0 = (rank + P * uk) * 16 + uu;                       // own unit pair j0, j0 + 1
    const bool uok = uk < nub && j0 < H;
    char* xbase = reinterpret_cast<char*>(a.cl_xchg) + (size_t)blk * CLB_BYTES;
    unsigned* xflags = reinterpret_cast<unsigned*>(xbase + 2 * CLB_G + CLB_C);
    const ClX xA = cl_x(xbase, CLB_G, xflags);                      // [16][200] {dr, dz, dn, dn*r} of layer 1
    const ClX xB = cl_x(xbase + CLB_G, CLB_G, xflags + 8);          // ... of layer 0
    const ClX xC = cl_x(xbase + 2 * CLB_G, CLB_C, xflags + 16);     // [16][300] total output gradient of the next step
    unsigned* tmo = a.cl_timeout;
    // a 16-byte piece of a gate-gradient buffer = 2 units x {dr, dz, dn, dn*r}: scatter into the two A operands
    constexpr int NPG = (TR * H / 2 + NTHR - 1) / NTHR;
    auto load_gates = [&](const ClX& x, bf16x8 (&pc)[NPG]) {
#pragma unroll
        for (int q = 0; q < NPG; ++q) { const int v = tid + q * NTHR; if (v < TR * H / 2) pc[q] = cl_load16(x, v * 16); }
    };
    auto put_gates = [&](const bf16x8 (&pc)[NPG]) {
#pragma unroll
        for (int q = 0; q < NPG; ++q) {
            const int v = tid + q * NTHR;
            if (v < TR * H / 2) {
                const int row = v / (H / 2), j = (v - row * (H / 2)) * 2;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dgi[row * LDGK + j + u] = pc[q][4 * u]; dgh[row * LDGK + j + u] = pc[q][4 * u];
                    dgi[row * LDGK + H + j + u] = pc[q][4 * u + 1]; dgh[row * LDGK + H + j + u] = pc[q][4 * u + 1];
                    dgi[row * LDGK + 2 * H + j + u] = pc[q][4 * u + 2]; dgh[row * LDGK + 2 * H + j + u] = pc[q][4 * u + 3];
                }
            }
        }
    };

    for (int i = tid; i < TR * LDX; i += NTHR) dob[i] = (bf16)0.f;
    for (int i = tid; i < 2 * TR * LDGK; i += NTHR) dgi[i] = (bf16)0.f;
    __syncthreads();
    // own embedding pairs of this thread: q-th pair = local pair pl + 32 q -> tile lt, columns e, e + 1
    int eo[2]; bool eok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lp = pl + 32 * q, lt = lp >> 3;
        eo[q] = (rank + P * lt) * 16 + 2 * (lp & 7);
        eok[q] = lt < noe && eo[q] < E;
    }
    float ws_sum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // step T-1: the total output gradient is the loss term alone, every rank reads all of it; the owner accounts for its columns
    for (int i = tid; i < TR * E; i += NTHR) {
        const int row = i / E, e = i - row * E;
        dob[row * LDX + e] = (bf16)(r0 + row < R ? a.dw[((size_t)(r0 + row) * T + (T - 1)) * E + e] : 0.f);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (eok[q] && gok) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float v = a.dw[(gr * T + (T - 1)) * E + eo[q] + c];
                ws_sum[q][c] += v;
                a.dout_b[((size_t)(T - 1) * R + gr) * EP + eo[q] + c] = (bf16)v;
            }
        }
    float dh0f[2] = {0.f, 0.f}, dh1f[2] = {0.f, 0.f};

    const __amdgpu_buffer_rsrc_t r_hoT = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_hoT), 0, 13 * KSX * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ih1T = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_ih1T), 0, 13 * KSG * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_hh1T = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_hh1T), 0, 13 * KSG * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_hh0T = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_hh0T), 0, 13 * KSG * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ih0T = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_ih0T), 0, 19 * KSG * 1024, 0x00020000);
    // a unit product: this wave's item = (tile li = wave % NUBMAX, plane kh = wave / NUBMAX)
    const int uli = wave % NUBMAX, ukh = wave / NUBMAX;
    const bool uitem = uli < nub;
    const int utile = rank + P * uli;
    auto load_pos = [&](bf16x8 (&dst)[KCH], int pos) {      // pos is a compile-time constant at every call site
        if (pos < 4) {
            if (!uitem) return;
            const int ks = pos == 0 ? KSX : KSG, kp = pos == 0 ? KPX : KPG;
            const __amdgpu_buffer_rsrc_t& rs = pos == 0 ? r_hoT : pos == 1 ? r_ih1T : pos == 2 ? r_hh1T : r_hh0T;
            const int k0 = ukh * kp, kc = min(kp, ks - k0);
            const int ub = (utile * ks + k0) * 1024;
#pragma unroll
            for (int s = 0; s < KCH; ++s)
                if (s < kc) dst[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ub + s * 1024, 0));
        } else {
            const int it = wave + NW * (pos - 4), li = it % NOEMAX, kh = it / NOEMAX;
            if (li >= noe || kh >= KH) return;
            const int k0 = kh * KPG, kc = min(KPG, KSG - k0);
            const int ub = ((rank + P * li) * KSG + k0) * 1024;
#pragma unroll
            for (int s = 0; s < KCH; ++s)
                if (s < kc) dst[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_ih0T, lane * 16, ub + s * 1024, 0));
        }
    };
    // consume chunk `pos` from ring slot pos % D: partial tile -> plane kh of `out`.  Chunks are requested explicitly
    // (request(pos)): the ones a phase starts with go out behind the loads of the exchange that precedes it, so that no
    // weight load is in flight while payload stores drain, and the stream travels while the exchange completes
    bf16x8 ring[D][KCH];
    auto request = [&](int pos, int slot) { load_pos(ring[slot], pos); };
    auto gemm_pos = [&](int pos, int slot, const bf16* A, int lda, float* out, int ldo) {
        int li, kh, k0, kc;
        bool ok;
        if (pos < 4) {
            const int ks = pos == 0 ? KSX : KSG, kp = pos == 0 ? KPX : KPG;
            li = uli; kh = ukh; k0 = kh * kp; kc = min(kp, ks - k0); ok = uitem && kc > 0;
        } else {
            const int it = wave + NW * (pos - 4);
            li = it % NOEMAX; kh = it / NOEMAX; k0 = kh * KPG; kc = min(KPG, KSG - k0); ok = li < noe && kh < KH && kc > 0;
        }
        if (ok) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KCH; ++s)
                if (s < kc) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + fr * lda + (k0 + s) * 32 + fq * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ring[slot][s], acc, 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(kh * TR + fq * 4 + j) * ldo + li * 16 + fr] = acc[j];
        }
    };
    static_assert(D == 2 && MTE == 2, "the explicit request schedule below is written for two ring slots and two feedback chunks");
    // Order of a step (chunk, ring slot):  hoT (0, s0) | gates 1, publish X1 | ih1T (1, s1) | gates 0, publish X2 | hh1T (2, s0)
    // DURING X2 | ih0T (4, s1) (5, s0) | output gradient of the next step, publish X3 | hh0T (3, s1) DURING X3 | carried
    // gradients += the two hidden-state products.  Those two products feed nothing but the carried gradients, so each of
    // them hides one exchange.
    request(0, 0);
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const unsigned ep = 3u * (unsigned)(T - 1 - t) + 1u;
        const bool first = t == 0;
        // saved gates of both layers for the own unit pair: requested before the products
        float s1[2][4], s0[2][4], hp1[2], hp0[2];
        unsigned short kpo = 0x0101;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = min(j0 + q, H - 1);
            const float* p1 = a.sav1 + ((size_t)t * R + gr) * 4 * H;
            const float* p0 = a.sav0 + ((size_t)t * R + gr) * 4 * H;
#pragma unroll
            for (int g = 0; g < 4; ++g) { s1[q][g] = p1[g * H + j]; s0[q][g] = p0[g * H + j]; }
            hp1[q] = a.h1_all[(size_t)t * RH + gr * H + j];
            hp0[q] = a.h0_all[(size_t)t * RH + gr * H + j];
        }
        if (KEEP && uok) kpo = gok ? *reinterpret_cast<const unsigned short*>(a.keep + (size_t)t * RH + gr * H + j0) : (unsigned short)0;
        // ---- dh1 += dOut * W_ho (own units)
        gemm_pos(0, 0, dob, LDX, o1, LDU);
        __syncthreads();
        if (uok) {
            bf16x8 pub;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = j0 + q, lc = uk * 16 + uu + q;
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
                if (gok) {
                    float o = 0.f;
#pragma unroll
                    for (int k = 0; k < KH; ++k) o += o1[(k * TR + grow) * LDU + lc];
                    const float r = s1[q][0], z = s1[q][1], n = s1[q][2], ghn = s1[q][3];
                    const float d = dh1f[q] + o;
                    dn = d * (1.0f - z) * (1.0f - n * n);
                    dz = d * (hp1[q] - n) * z * (1.0f - z);
                    dr = dn * ghn * r * (1.0f - r);
                    dnr = dn * r;
                    dd = d * z;
                    bf16* gi = a.dgi1_b + ((size_t)t * R + gr) * GP;
                    bf16* gh = a.dgh1_b + ((size_t)t * R + gr) * GP;
                    gi[j] = (bf16)dr; gi[H + j] = (bf16)dz; gi[2 * H + j] = (bf16)dn;
                    gh[j] = (bf16)dr; gh[H + j] = (bf16)dz; gh[2 * H + j] = (bf16)dnr;
                }
                dh1f[q] = dd;
                pub[4 * q] = (bf16)dr; pub[4 * q + 1] = (bf16)dz; pub[4 * q + 2] = (bf16)dn; pub[4 * q + 3] = (bf16)dnr;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4c, pub), xA.rs, (grow * H + j0) * 8, 0, 16);
        }
        cl_signal(xA, rank, ep, tid);
        cl_wait<P>(xA, ep, tmo, tid);
        {
            bf16x8 pc[NPG];
            load_gates(xA, pc);
            request(1, 1);
            put_gates(pc);
        }
        __syncthreads();
        // ---- dmid = dgi1 * W_ih1 ; dh1[t-1] += dgh1 * W_hh1 (own units)
        gemm_pos(1, 1, dgi, LDGK, o1, LDU);
        __syncthreads();
        if (uok) {
            bf16x8 pub;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = j0 + q, lc = uk * 16 + uu + q;
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
                float om = 0.f;
#pragma unroll
                for (int k = 0; k < KH; ++k) om += o1[(k * TR + grow) * LDU + lc];
                if (gok) {
                    const float r = s0[q][0], z = s0[q][1], n = s0[q][2], ghn = s0[q][3];
                    float dm = om;
                    if (KEEP) dm = ((kpo >> (8 * q)) & 0xff) ? dm * a.keep_scale : 0.f;
                    const float d = dh0f[q] + dm;
                    dn = d * (1.0f - z) * (1.0f - n * n);
                    dz = d * (hp0[q] - n) * z * (1.0f - z);
                    dr = dn * ghn * r * (1.0f - r);
                    dnr = dn * r;
                    dd = d * z;
                    bf16* gi = a.dgi0_b + ((size_t)t * R + gr) * GP;
                    bf16* gh = a.dgh0_b + ((size_t)t * R + gr) * GP;
                    gi[j] = (bf16)dr; gi[H + j] = (bf16)dz; gi[2 * H + j] = (bf16)dn;
                    gh[j] = (bf16)dr; gh[H + j] = (bf16)dz; gh[2 * H + j] = (bf16)dnr;
                }
                dh0f[q] = dd;
                pub[4 * q] = (bf16)dr; pub[4 * q + 1] = (bf16)dz; pub[4 * q + 2] = (bf16)dn; pub[4 * q + 3] = (bf16)dnr;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4c, pub), xB.rs, (grow * H + j0) * 8, 0, 16);
        }
        cl_signal(xB, rank, ep + 1, tid);
        // dh1[t-1] += dgh1 * W_hh1 (own units): still layer 1's gate gradients in dgh -- it runs while exchange X2 completes
        request(2, 0);
        gemm_pos(2, 0, dgh, LDGK, o3, LDU);
        cl_wait<P>(xB, ep + 1, tmo, tid);           // (its barrier: every wave is done with the old dgh)
        {
            bf16x8 pc[NPG];
            load_gates(xB, pc);
            request(4, 1); request(5, 0);
            put_gates(pc);
        }
        __syncthreads();
        // ---- feedback into the previous output = dgi0 * W_ih0[:, :300] (own embedding tiles)
        gemm_pos(4, 1, dgi, LDGK, fbp, LDE);
        gemm_pos(5, 0, dgi, LDGK, fbp, LDE);
        __syncthreads();
        if (first) {                                // last step of the loop: no exchange left to hide the product behind
            request(3, 1);
            gemm_pos(3, 1, dgh, LDGK, o2, LDU);
        }
        if (!first) {
            // total output gradient of step t-1 for the own columns: loss term + feedback; accounted, saved, published
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (eok[q]) {
                    const int lp = pl + 32 * q, lt = lp >> 3, lcol = lt * 16 + 2 * (lp & 7);
                    float v[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float f = 0.f;
#pragma unroll
                        for (int k = 0; k < KH; ++k) f += fbp[(k * TR + grow) * LDE + lcol + c];
                        v[c] = (gok ? a.dw[(gr * T + (t - 1)) * E + eo[q] + c] : 0.f) + f;
                        if (gok) { ws_sum[q][c] += v[c]; a.dout_b[((size_t)(t - 1) * R + gr) * EP + eo[q] + c] = (bf16)v[c]; }
                    }
                    cl_store4(xC, (grow * E + eo[q]) * 2, (bf16)v[0], (bf16)v[1]);
                }
            cl_signal(xC, rank, ep + 2, tid);
            // dh0[t-1] += dgh0 * W_hh0 (own units) runs while exchange X3 completes
            request(3, 1);
            gemm_pos(3, 1, dgh, LDGK, o2, LDU);
            cl_wait<P>(xC, ep + 2, tmo, tid);
            constexpr int NPC = (TR * E / 4 + NTHR - 1) / NTHR;     // 8 bytes = 4 columns
            u32x2 x[NPC];
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int v = tid + q * NTHR;
                if (v < TR * E / 4) { const int row = v / (E / 4), e = (v - row * (E / 4)) * 4; x[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xC.rs, (row * E + e) * 2, 0, 16)); }
            }
            request(0, 0);
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int v = tid + q * NTHR;
                if (v < TR * E / 4) { const int row = v / (E / 4), e = (v - row * (E / 4)) * 4; *reinterpret_cast<u32x2*>(dob + row * LDX + e) = x[q]; }
            }
        }
        __syncthreads();
        if (uok && gok) {                           // carried gradients += the two hidden-state products of this step
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float oh0 = 0.f, oh1 = 0.f;
#pragma unroll
                for (int k = 0; k < KH; ++k) { oh0 += o2[(k * TR + grow) * LDU + uk * 16 + uu + q]; oh1 += o3[(k * TR + grow) * LDU + uk * 16 + uu + q]; }
                dh0f[q] += oh0; dh1f[q] += oh1;
            }
        }
    }
    if (uok && gok) {
#pragma unroll
        for (int q = 0; q < 2; ++q) a.dhinit[gr * H + j0 + q] = dh0f[q] + dh1f[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (eok[q] && gok) { a.dwsum[gr * E + eo[q]] = ws_sum[q][0]; a.dwsum[gr * E + eo[q] + 1] = ws_sum[q][1]; }
}

// ================================================================== backward (BPTT), COMPOSED cluster form (8 ranks, resident)
// Mirror of coco_dec_fwd_c8_kernel.  The gradient entering h1[t] through the output is dOut[t] W_ho with
// dOut[t] = dw[t] + dgi0[t+1] W_ih0x; composed: dw[t] W_ho (no recurrence in it: its operand comes straight from the bf16 copy
// of the loss gradient the MSE kernel leaves, a step ahead) + dgi0[t+1] W_comb (reads the layer-0 gate gradients every rank
// holds after exchange X2).  The all-gather of the total output gradient disappears, and so does everything that only needs
// dOut: W_ho's weight gradient and the z-term of the output projection get it from ONE batched GEMM over the saved dgi0
// afterwards (coco_text.hip), the kernel only keeps the time sum of dw for its own embedding columns.  What is left is the
// recurrence proper -- chain of a step: gates 1, publish X1 | (hh0T of the step before, during X1) | ih1T | gates 0, publish
// X2 | (hh1T and hoT dw[t-1], during X2) | combT -- with every weight fragment resident: 4 x 5 + 3 k-steps x 4 VGPRs a wave.
template <bool KEEP>
__global__ __launch_bounds__(NTHR) void coco_dec_bwd_c8_kernel(const CocoDecBwdArgs a) {
    constexpr int P = 8, NUBMAX = 2, KH = NW / NUBMAX;
    constexpr int KSG = GP / 32, KSX = XP / 32;                       // 19, 10 k-steps
    constexpr int KPG = (KSG + KH - 1) / KH, KPX = (KSX + KH - 1) / KH;   // 5, 3 k-steps per plane
    constexpr int LDU = NUBMAX * 16 + 4, PLU = KH * TR * LDU;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* o1 = reinterpret_cast<float*>(smem);                     // [KH][16][LDU] dgi1 W_ih1
    float* o2 = o1 + PLU;                                           // dgh0 W_hh0 (of the step before)
    float* o3 = o2 + PLU;                                           // dgh1 W_hh1 + dw W_ho
    float* oc = o3 + PLU;                                           // dgi0 W_comb
    bf16* dgi = reinterpret_cast<bf16*>(oc + PLU);                  // [16][LDGK]
    bf16* dgh = dgi + TR * LDGK;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nblk = (a.R + TR - 1) / TR, nblk_pad = (nblk + 7) / 8 * 8;
    const int rank = blockIdx.x / nblk_pad, blk = blockIdx.x - rank * nblk_pad;
    if (blk >= nblk) return;
    const int r0 = blk * TR, R = a.R, T = a.T;
    const int nub = (13 - rank + P - 1) / P, noe = (19 - rank + P - 1) / P;
    const int grow = tid >> 5, pl = tid & 31;
    const bool gok = r0 + grow < R;
    const int gr = gok ? r0 + grow : 0;
    // ONE own unit per thread (a rank has at most 32): local column pl of the planes
    const int uk = pl >> 4, ju = (rank + P * uk) * 16 + (pl & 15);
    const bool uok = uk < nub && ju < H, act = uok && gok;
    // own embedding columns (time sum of the loss gradient): pair pl of the rank's noe * 8
    const int eo = (rank + P * (pl >> 3)) * 16 + 2 * (pl & 7);
    const bool eact = (pl >> 3) < noe && eo < E && gok;
    char* xbase = reinterpret_cast<char*>(a.cl_xchg) + (size_t)blk * CLB_BYTES;
    unsigned* xflags = reinterpret_cast<unsigned*>(xbase + 2 * CLB_G + CLB_C);
    const ClX xA = cl_x(xbase, CLB_G, xflags);                      // [16][200] {dr, dz, dn, dn*r} of layer 1
    const ClX xB = cl_x(xbase + CLB_G, CLB_G, xflags + 8);          // ... of layer 0
    unsigned* tmo = a.cl_timeout;
    // every per-step global access goes through a buffer descriptor: per-thread byte offset (loop-invariant, 32 bits) + a scalar
    // offset for the step (64-bit addresses per access cost two registers each)
    auto mkrs = [](const void* p, size_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const size_t TRn = (size_t)T * R;
    const __amdgpu_buffer_rsrc_t q_dw = mkrs(a.dw, TRn * E * 4), q_dw16 = mkrs(a.dw16, TRn * XP * 2),
                                 q_sav0 = mkrs(a.sav0, TRn * 4 * H * 4), q_sav1 = mkrs(a.sav1, TRn * 4 * H * 4),
                                 q_h0 = mkrs(a.h0_all, (TRn + R) * H * 4), q_h1 = mkrs(a.h1_all, (TRn + R) * H * 4),
                                 q_gi0 = mkrs(a.dgi0_b, TRn * GP * 2), q_gh0 = mkrs(a.dgh0_b, TRn * GP * 2),
                                 q_gi1 = mkrs(a.dgi1_b, TRn * GP * 2), q_gh1 = mkrs(a.dgh1_b, TRn * GP * 2);
    const int vo_g = (gr * GP + ju) * 2, vo_s = (gr * 4 * H + ju) * 4, vo_h = (gr * H + ju) * 4;
    auto ldf = [](const __amdgpu_buffer_rsrc_t& rs, int vo, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0)); };
    auto stb = [](const __amdgpu_buffer_rsrc_t& rs, int vo, int so, float x) {
        const bf16 b = (bf16)x;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, b), rs, vo, so, 0);
    };
    // gate gradients of the own unit, step t: gi = (dr, dz, dn), gh = (dr, dz, dn r)
    auto put_gv = [&](const __amdgpu_buffer_rsrc_t& qi, const __amdgpu_buffer_rsrc_t& qh, int t, const float (&gv)[4]) {
        const int so = t * R * GP * 2;
        stb(qi, vo_g, so, gv[0]); stb(qi, vo_g + H * 2, so, gv[1]); stb(qi, vo_g + 2 * H * 2, so, gv[2]);
        stb(qh, vo_g, so, gv[0]); stb(qh, vo_g + H * 2, so, gv[1]); stb(qh, vo_g + 2 * H * 2, so, gv[3]);
    };
    // saved gates (r, z, n, W_hn h + b_hn) and previous hidden state of the own unit, step t
    auto get_sav = [&](const __amdgpu_buffer_rsrc_t& qs, const __amdgpu_buffer_rsrc_t& qh, int t, float (&sv)[4], float& hp) {
        const int so = t * R * 4 * H * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) sv[g] = ldf(qs, vo_s + g * H * 4, so);
        hp = ldf(qh, vo_h, t * R * H * 4);
    };
    constexpr int NPG = (TR * H / 2 + NTHR - 1) / NTHR;
    auto load_gates = [&](const ClX& x, bf16x8 (&pc)[NPG]) {
#pragma unroll
        for (int q = 0; q < NPG; ++q) { const int v = tid + q * NTHR; if (v < TR * H / 2) pc[q] = cl_load16(x, v * 16); }
    };
    auto put_gates = [&](const bf16x8 (&pc)[NPG]) {       // a 16-byte piece = 2 units x {dr, dz, dn, dn*r}: into the two A operands
#pragma unroll
        for (int q = 0; q < NPG; ++q) {
            const int v = tid + q * NTHR;
            if (v < TR * H / 2) {
                const int row = v / (H / 2), j = (v - row * (H / 2)) * 2;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dgi[row * LDGK + j + u] = pc[q][4 * u]; dgh[row * LDGK + j + u] = pc[q][4 * u];
                    dgi[row * LDGK + H + j + u] = pc[q][4 * u + 1]; dgh[row * LDGK + H + j + u] = pc[q][4 * u + 1];
                    dgi[row * LDGK + 2 * H + j + u] = pc[q][4 * u + 2]; dgh[row * LDGK + 2 * H + j + u] = pc[q][4 * u + 3];
                }
            }
        }
    };
    for (int i = tid; i < 2 * TR * LDGK; i += NTHR) dgi[i] = (bf16)0.f;
    for (int i = tid; i < 4 * PLU; i += NTHR) o1[i] = 0.f;
    float ws_sum[2] = {0.f, 0.f};
    float gsum[3] = {0.f, 0.f, 0.f};        // time sum of the own unit's layer-0 input-side gate gradients (as stored: bf16 values)
    float dh0f = 0.f, dh1f = 0.f;
    // ---- resident weights: this wave's item of a unit product = (tile uli = wave % 2, K plane ukh = wave / 2)
    const int uli = wave % NUBMAX, ukh = wave / NUBMAX;
    const bool uitem = uli < nub;
    const int kcx = uitem ? max(0, min(KPX, KSX - ukh * KPX)) : 0, kcg = uitem ? max(0, min(KPG, KSG - ukh * KPG)) : 0;
    bf16x8 w_ho[KPX], w_c[KPG], w_ih1[KPG], w_hh1[KPG], w_hh0[KPG];
    {
        auto rs = [](const bf16* p, int tiles, int ks) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p), 0, tiles * ks * 1024, 0x00020000); };
        const __amdgpu_buffer_rsrc_t r_ho = rs(a.w_hoT, 13, KSX), r_c = rs(a.w_combT, 13, KSG), r_1 = rs(a.w_ih1T, 13, KSG), r_2 = rs(a.w_hh1T, 13, KSG),
                                     r_0 = rs(a.w_hh0T, 13, KSG);
        const int utile = uitem ? rank + P * uli : 0;
        const bf16x8 z8 = {};
#pragma unroll
        for (int s = 0; s < KPX; ++s)
            w_ho[s] = s < kcx ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_ho, lane * 16, (utile * KSX + ukh * KPX + s) * 1024, 0)) : z8;
#pragma unroll
        for (int s = 0; s < KPG; ++s) {
            const int ub = (utile * KSG + ukh * KPG + s) * 1024;
            const bool ok = s < kcg;
            w_c[s] = ok ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_c, lane * 16, ub, 0)) : z8;
            w_ih1[s] = ok ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_1, lane * 16, ub, 0)) : z8;
            w_hh1[s] = ok ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_2, lane * 16, ub, 0)) : z8;
            w_hh0[s] = ok ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r_0, lane * 16, ub, 0)) : z8;
        }
    }
    // A fragments of dw[t] W_ho straight from the bf16 copy of the loss gradient ([row][t][320], pad columns zero): row fr of
    // the block, this wave's K plane
    const int vo_a = ((min(r0 + fr, R - 1) * T) * XP + ukh * KPX * 32 + fq * 8) * 2;
    bf16x8 dwa[KPX];
    auto load_dwa = [&](int t) {
#pragma unroll
        for (int s = 0; s < KPX; ++s)
            if (s < kcx) dwa[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(q_dw16, vo_a + s * 64, t * XP * 2, 0));
    };
    // partial tile of this wave's item: k-steps [k0, k0 + kc) of the LDS operand against resident fragments, on top of `acc`
    auto umma = [&](const auto& w, const bf16* A, f32x4 acc) {
        constexpr int N = sizeof(w) / sizeof(w[0]);
#pragma unroll
        for (int s = 0; s < N; ++s)
            if (s < kcg) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + fr * LDGK + (ukh * KPG + s) * 32 + fq * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, w[s], acc, 0, 0, 0);
            }
        return acc;
    };
    auto uput = [&](float* out, f32x4 acc) {
        if (kcg > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(ukh * TR + fq * 4 + j) * LDU + uli * 16 + fr] = acc[j];
        }
    };
    auto homma = [&](f32x4 acc) {           // += dw W_ho, K plane of this wave (fragments in dwa)
#pragma unroll
        for (int s = 0; s < KPX; ++s)
            if (s < kcx) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dwa[s], w_ho[s], acc, 0, 0, 0);
        return acc;
    };
    auto usum = [&](const float* o) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < KH; ++k) t += o[(k * TR + grow) * LDU + pl];
        return t;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, hp1 = 0.f;
    if (act) get_sav(q_sav1, q_h1, T - 1, s1, hp1);
    load_dwa(T - 1);
    __syncthreads();
    uput(o3, homma(zero4));
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const unsigned ep = (unsigned)(T - t);
        const bool newest = t == T - 1, first = t == 0;
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, hp0 = 0.f, dwo[2] = {0.f, 0.f};
        unsigned kp = 1;
        // ---- layer-1 gates backward: dh1 = carried + (dw W_ho + dgh1[t+1] W_hh1) + dgi0[t+1] W_comb
        {
            float gv[4] = {0.f, 0.f, 0.f, 0.f};
            if (uok) {
                float dd = 0.f;
                if (gok) {
                    const float r = s1[0], z = s1[1], n = s1[2], ghn = s1[3];
                    const float d = dh1f + usum(o3) + usum(oc);
                    gv[2] = d * (1.0f - z) * (1.0f - n * n);
                    gv[1] = d * (hp1 - n) * z * (1.0f - z);
                    gv[0] = gv[2] * ghn * r * (1.0f - r);
                    gv[3] = gv[2] * r;
                    dd = d * z;
                }
                dh1f = dd;
                cl_store8(xA, (grow * H + ju) * 8, (bf16)gv[0], (bf16)gv[1], (bf16)gv[2], (bf16)gv[3]);
            }
            cl_signal(xA, rank, ep, tid);
            if (act) put_gv(q_gi1, q_gh1, t, gv);       // (behind the flag: the exchange does not wait for these)
        }
        // requests that have the window of X1 to arrive: saved gates of layer 0, the operand of the next dw W_ho, own columns of dw
        if (act) get_sav(q_sav0, q_h0, t, s0, hp0);
        if (KEEP && act) kp = a.keep[(size_t)t * R * H + (size_t)gr * H + ju];
        if (!first) load_dwa(t - 1);
        if (eact) { dwo[0] = ldf(q_dw, (gr * T * E + eo) * 4, t * E * 4); dwo[1] = ldf(q_dw, (gr * T * E + eo) * 4 + 4, t * E * 4); }
        // during X1 (dgh still holds layer 0's gate gradients of step t+1): the dh0 carry
        if (!newest) uput(o2, umma(w_hh0, dgh, zero4));
        cl_wait<P>(xA, ep, tmo, tid);
        {
            bf16x8 pc[NPG];
            load_gates(xA, pc);
            ws_sum[0] += dwo[0]; ws_sum[1] += dwo[1];
            put_gates(pc);
        }
        __syncthreads();
        // ---- dmid = dgi1 W_ih1 (own units)
        uput(o1, umma(w_ih1, dgi, zero4));
        __syncthreads();
        {
            float gv[4] = {0.f, 0.f, 0.f, 0.f};
            if (uok) {
                float dd = 0.f;
                if (gok) {
                    const float r = s0[0], z = s0[1], n = s0[2], ghn = s0[3];
                    float dm = usum(o1);
                    if (KEEP) dm = kp ? dm * a.keep_scale : 0.f;
                    const float d = dh0f + usum(o2) + dm;
                    gv[2] = d * (1.0f - z) * (1.0f - n * n);
                    gv[1] = d * (hp0 - n) * z * (1.0f - z);
                    gv[0] = gv[2] * ghn * r * (1.0f - r);
                    gv[3] = gv[2] * r;
                    dd = d * z;
                }
                dh0f = dd;
                cl_store8(xB, (grow * H + ju) * 8, (bf16)gv[0], (bf16)gv[1], (bf16)gv[2], (bf16)gv[3]);
            }
            cl_signal(xB, rank, ep, tid);
            if (act) put_gv(q_gi0, q_gh0, t, gv);
            if (!first) { gsum[0] += (float)(bf16)gv[0]; gsum[1] += (float)(bf16)gv[1]; gsum[2] += (float)(bf16)gv[2]; }
            else if (act) {     // last step of the loop: the sums over t >= 1 (what the output projection's z-term sees) and over all t
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    a.dzi1[(size_t)gr * G + g * H + ju] = gsum[g];
                    a.dzi0[(size_t)gr * G + g * H + ju] = gsum[g] + (float)(bf16)gv[g];
                }
            }
        }
        if (!first && act) get_sav(q_sav1, q_h1, t - 1, s1, hp1);
        // during X2 (dgh still holds layer 1's gate gradients): the dh1 carry, and the loss term of the step below on top of it
        {
            f32x4 acc = umma(w_hh1, dgh, zero4);
            if (!first) acc = homma(acc);
            uput(o3, acc);
        }
        cl_wait<P>(xB, ep, tmo, tid);
        {
            bf16x8 pc[NPG];
            load_gates(xB, pc);
            put_gates(pc);
        }
        __syncthreads();
        if (!first) uput(oc, umma(w_c, dgi, zero4));
        __syncthreads();
    }
    // what the loop hides inside the next step's first window: dgh0[0] W_hh0
    uput(o2, umma(w_hh0, dgh, zero4));
    __syncthreads();
    if (act) a.dhinit[(size_t)gr * H + ju] = dh0f + usum(o2) + dh1f + usum(o3);
    if (eact) { a.dwsum[(size_t)gr * E + eo] = ws_sum[0]; a.dwsum[(size_t)gr * E + eo + 1] = ws_sum[1]; }
}

// ================================================================== caption encoder (forward direction of the bi-GRU)
// coco/model.py:236-245.  The input projection of all T steps is one batched GEMM done by the caller; what is sequential is
// h[t] = GRU(gi[t], h[t-1]): one hidden projection (W_hh, 272 KB bf16) and the gate math per step.  Same scheme as the
// decoder: 16 rows per workgroup for the whole recurrence, the weights streamed through the register ring.
template <bool SAVE>
__global__ __launch_bounds__(NTHR) void coco_enc_fwd_kernel(const CocoEncFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gb = reinterpret_cast<float*>(smem);                     // [16][LDG] hidden projection
    float* hf = gb + TR * LDG;                                      // [16][H]
    float* bias = hf + TR * H;                                      // b_hh [G]
    bf16* hb = reinterpret_cast<bf16*>(bias + G);                   // [16][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * TR, R = a.B, T = a.T;
    const size_t RH = (size_t)R * H;
    const int grow = tid >> 5, c0 = tid & 31;
    const bool gok = r0 + grow < R;
    const size_t gr = gok ? r0 + grow : 0;
    for (int i = tid; i < TR * H; i += NTHR) hf[i] = 0.f;
    for (int i = tid; i < TR * LDH; i += NTHR) hb[i] = (bf16)0.f;
    for (int i = tid; i < G; i += NTHR) bias[i] = a.bhh[i];
    if (SAVE && gok) {          // slice 0 of the saved operand: h before the first step (zeros) + the 1.0 column of the bias gradient
#pragma unroll
        for (int q = 0; q < NQH; ++q)
            if (c0 + 32 * q < H) a.hb_all[gr * HP + c0 + 32 * q] = (bf16)0.f;
        if (c0 == 0) a.hb_all[gr * HP + H] = (bf16)1.f;
    }
    // chunk schedule of a step (ring depth 3): hh 5 tile slots + 1 dummy = 6 chunks
    constexpr int D = 3;
    const WMat<HP / 32, GP / 16> d_hh(a.w_hh);
    bf16x8 ring[D][KCH];
#pragma unroll
    for (int q = 0; q < D; ++q) load_chunk(ring[q], d_hh, q, wave, lane);
    __syncthreads();
    const float* pb = gb + grow * LDG;
    for (int t = 0; t < T; ++t) {
        float g_r[NQH], g_z[NQH], g_n[NQH];
        const float* gi = a.gi + (gr * T + t) * G;
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = min(c0 + 32 * q, H - 1);
            g_r[q] = gi[j]; g_z[q] = gi[H + j]; g_n[q] = gi[2 * H + j];
        }
        stream_gemm<6, D, 0>(hb, LDH, d_hh, gb, LDG, ring, d_hh, t + 1 < T, wave, lane);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = c0 + 32 * q;
            if (j < H) {
                const int i = grow * H + j;
                const float r = sigm(g_r[q] + pb[j] + bias[j]);
                const float z = sigm(g_z[q] + pb[H + j] + bias[H + j]);
                const float ghn = pb[2 * H + j] + bias[2 * H + j];
                const float n = tanh_fast(g_n[q] + r * ghn);
                const float hn = (1.0f - z) * n + z * hf[i];
                hf[i] = hn;
                hb[grow * LDH + j] = (bf16)hn;
                if (SAVE && q == 0 && c0 == 0 && gok && t + 1 < T) a.hb_all[((size_t)(t + 1) * R + gr) * HP + H] = (bf16)1.f;
                if (gok) {
                    a.h_all[(size_t)t * RH + gr * H + j] = hn;
                    if (SAVE) {
                        float* s = a.sav + ((size_t)t * R + gr) * 4 * H;
                        s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
                        if (t + 1 < T) a.hb_all[((size_t)(t + 1) * R + gr) * HP + j] = (bf16)hn;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// The same recurrence with W_hh RESIDENT on the CU: 272 KB of bf16 fit the registers of 8 waves (84 VGPRs per lane for a
// wave's first 16-unit block of the three gates) plus 105 KB of LDS (the second block of waves 0-4), so a step streams
// nothing.  The weights are packed per gate (three fragment-major [208][224] matrices), a wave owns unit blocks
// wave and wave + 8 of ALL THREE gates, so r, z and n of a hidden unit meet in the accumulators of one lane (MFMA C layout:
// rows 4*(lane/16) .. +3 of column lane%16) and the gate math runs in registers -- the state h lives there too, in fp32.
// What a lane owns is 4 consecutive ROWS of one unit, so the per-step operands and saves of this path are laid out with
// the batch row as the fastest index ([gate*H + unit][t][B] for the input projection, [t][gate][unit][B] for the saves): one
// 16-byte load / store per (gate, unit) instead of four
// scattered ones.  The bf16 copy of h (next step's A operand, and the row-major operand of the batched weight gradient)
// is double-buffered in LDS and written out by all threads as 16-byte vectors one step later.  One barrier per step.
template <bool SAVE>
__global__ __launch_bounds__(NTHR) void coco_enc_fwd_res_kernel(const CocoEncFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* hb = reinterpret_cast<bf16*>(smem);                       // [2][16][LDH]
    bf16* wl = hb + 2 * TR * LDH;                                   // [5 waves][3 gates][7 k-steps][64 lanes][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int r0 = blockIdx.x * TR, R = a.B, T = a.T;
    constexpr int KS = HP / 32, NB = (H + 15) / 16;                 // 7 k-steps, 13 unit blocks
    constexpr size_t GATE = (size_t)NB * 16 * HP;                   // elements of one packed gate matrix
    const bool two = wave + NW < NB;                                // waves 0-4 own a second block
    bf16x8 w0[3][KS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            w0[g][s] = *reinterpret_cast<const bf16x8*>(a.w_hh + g * GATE + ((size_t)(wave * KS + s) * 64 + lane) * 8);
            if (two) *reinterpret_cast<bf16x8*>(wl + ((size_t)((wave * 3 + g) * KS + s) * 64 + lane) * 8) =
                *reinterpret_cast<const bf16x8*>(a.w_hh + g * GATE + ((size_t)((wave + NW) * KS + s) * 64 + lane) * 8);
        }
    for (int i = tid; i < 2 * TR * LDH; i += NTHR) hb[i] = (bf16)0.f;
    int ju[2]; bool uok[2]; float br[2], bz[2], bn[2], bin[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ju[i] = (wave + NW * i) * 16 + fr;
        uok[i] = (i == 0 || two) && ju[i] < H;
        const int j = min(ju[i], H - 1);
        br[i] = a.bhh[j] + a.bih[j]; bz[i] = a.bhh[H + j] + a.bih[H + j]; bn[i] = a.bhh[2 * H + j]; bin[i] = a.bih[2 * H + j];
    }
    const int row4 = r0 + fq * 4;                                   // this lane's 4 rows (B is a multiple of 4: all or none exist)
    const bool rok = row4 < R;
    const size_t rbase = rok ? row4 : 0;
    // copy of a finished 16-row bf16 state tile to the [t][row] operand of the weight gradient: thread -> (row, 16-byte vector)
    const int crow = tid / (HP / 8), cvec = tid - crow * (HP / 8);
    auto put_rows = [&](const bf16* src, int slice) {
        if (tid < TR * (HP / 8) && r0 + crow < R) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(src + crow * LDH + cvec * 8);
            if (cvec == H / 8) v[0] = (bf16)1.f;                    // column H = 1.0: the bias gradient rides in the weight gradient
            *reinterpret_cast<bf16x8*>(a.hb_all + ((size_t)slice * R + r0 + crow) * HP + cvec * 8) = v;
        }
    };
    float hst[2][4] = {};
    // the input projection of a step, [gate*H + unit][t][B] (no bias), is requested a whole step ahead: it comes from HBM /
    // Infinity Cache (31 MB per pass), and one MFMA phase does not cover that latency
    f32x4 gi[2][3], gnx[2][3];
    auto fetch = [&](f32x4 (&dst)[2][3], int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                dst[i][g] = *reinterpret_cast<const f32x4*>(a.gi + ((size_t)(g * H + min(ju[i], H - 1)) * T + t) * R + rbase);
    };
    fetch(gnx, 0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const bf16* hc = hb + (t & 1) * TR * LDH;
        bf16* hn_b = hb + ((t + 1) & 1) * TR * LDH;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[i][g] = gnx[i][g];
        if (t + 1 < T) fetch(gnx, t + 1);
        if (SAVE) put_rows(hc, t);                                  // h BEFORE this step (slice 0 = zeros)
        bf16x8 af[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const bf16x8*>(hc + fr * LDH + s * 32 + fq * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !two) break;
            f32x4 acc[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 wv = i == 0 ? w0[g][s] : *reinterpret_cast<const bf16x8*>(wl + ((size_t)((wave * 3 + g) * KS + s) * 64 + lane) * 8);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], wv, acc[g], 0, 0, 0);
                }
            }
            if (uok[i]) {
                const int j = ju[i];
                f32x4 vr, vz, vn, vg, vh;
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) {
                    const float r = sigm(gi[i][0][jr] + acc[0][jr] + br[i]);
                    const float z = sigm(gi[i][1][jr] + acc[1][jr] + bz[i]);
                    const float ghn = acc[2][jr] + bn[i];
                    const float n = tanh_fast(gi[i][2][jr] + bin[i] + r * ghn);
                    const float hn = (1.0f - z) * n + z * hst[i][jr];
                    hst[i][jr] = hn;
                    hn_b[(fq * 4 + jr) * LDH + j] = (bf16)(rok ? hn : 0.f);
                    vr[jr] = r; vz[jr] = z; vn[jr] = n; vg[jr] = ghn; vh[jr] = hn;
                    if (rok && t + 1 == T) a.h_last[(rbase + jr) * H + j] = hn;
                }
                if (SAVE && rok) {
                    float* sv = a.sav + ((size_t)t * 4 * H + j) * R + rbase;          // [t][gate][unit][B]
                    *reinterpret_cast<f32x4*>(sv) = vr;
                    *reinterpret_cast<f32x4*>(sv + (size_t)H * R) = vz;
                    *reinterpret_cast<f32x4*>(sv + (size_t)2 * H * R) = vn;
                    *reinterpret_cast<f32x4*>(sv + (size_t)3 * H * R) = vg;
                    *reinterpret_cast<f32x4*>(a.h_all + ((size_t)t * H + j) * R + rbase) = vh;    // [t][unit][B]
                }
            }
        }
        __syncthreads();
    }
}

// BPTT with W_hh^T resident (253 KB: a wave's first 16-unit tile in 76 VGPRs, the second one of waves 0-4 in 97 KB of
// LDS).  The product dgh[t] * W_hh lands in the MFMA C layout, i.e. with the lane that owns (4 rows, unit j) of the gate
// backward: the carried gradient never leaves registers.  dgh / dgi are written to LDS row-major (the next product's A
// operand) and copied from there to the [t][row] operands of the batched weight gradients with 16-byte stores.
__global__ __launch_bounds__(NTHR) void coco_enc_bwd_res_kernel(const CocoEncBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* dgb = reinterpret_cast<bf16*>(smem);                      // [2][16][LDGK] dgh (r | z | n*r)
    bf16* dnb = dgb + 2 * TR * LDGK;                                // [2][16][LDH]  dn (the n-columns of dgi)
    bf16* wl = dnb + 2 * TR * LDH;                                  // [5 waves][19 k-steps][64 lanes][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int r0 = blockIdx.x * TR, R = a.B, T = a.T;
    constexpr int KS = GP / 32, NB = (H + 15) / 16;                 // 19 k-steps, 13 unit tiles
    const bool two = wave + NW < NB;
    bf16x8 w0[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        w0[s] = *reinterpret_cast<const bf16x8*>(a.w_hhT + ((size_t)(wave * KS + s) * 64 + lane) * 8);
        if (two) *reinterpret_cast<bf16x8*>(wl + ((size_t)(wave * KS + s) * 64 + lane) * 8) =
            *reinterpret_cast<const bf16x8*>(a.w_hhT + ((size_t)((wave + NW) * KS + s) * 64 + lane) * 8);
    }
    for (int i = tid; i < 2 * TR * LDGK + 2 * TR * LDH; i += NTHR) dgb[i] = (bf16)0.f;
    int ju[2]; bool uok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ju[i] = (wave + NW * i) * 16 + fr; uok[i] = (i == 0 || two) && ju[i] < H; }
    const int row4 = r0 + fq * 4;
    const bool rok = row4 < R;
    const size_t rbase = rok ? row4 : 0;
    float dcar[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) dcar[i][jr] = rok && uok[i] ? a.dh_init[(rbase + jr) * H + ju[i]] : 0.f;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // saved gates of a step and h[t-1] ([..][unit][B]) are requested a whole step ahead (HBM / Infinity Cache latency)
    f32x4 sv[2][4], hp[2], svn[2][4], hpn[2];
    auto fetch = [&](f32x4 (&s4)[2][4], f32x4 (&h4)[2], int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = min(ju[i], H - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) s4[i][g] = *reinterpret_cast<const f32x4*>(a.sav + ((size_t)(t * 4 + g) * H + j) * R + rbase);
            h4[i] = t > 0 ? *reinterpret_cast<const f32x4*>(a.h_all + ((size_t)(t - 1) * H + j) * R + rbase) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(svn, hpn, T - 1);
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        bf16* dg = dgb + (t & 1) * TR * LDGK;
        bf16* dn_s = dnb + (t & 1) * TR * LDH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            hp[i] = hpn[i];
#pragma unroll
            for (int g = 0; g < 4; ++g) sv[i][g] = svn[i][g];
        }
        if (t > 0) fetch(svn, hpn, t - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!uok[i]) continue;
            const int j = ju[i];
#pragma unroll
            for (int jr = 0; jr < 4; ++jr) {
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
                if (rok) {
                    const float r = sv[i][0][jr], z = sv[i][1][jr], n = sv[i][2][jr], ghn = sv[i][3][jr];
                    const float d = dcar[i][jr] + acc[i][jr];
                    dn = d * (1.0f - z) * (1.0f - n * n);
                    dz = d * (hp[i][jr] - n) * z * (1.0f - z);
                    dr = dn * ghn * r * (1.0f - r);
                    dnr = dn * r;
                    dd = d * z;
                }
                dcar[i][jr] = dd;
                bf16* p = dg + (fq * 4 + jr) * LDGK;
                p[j] = (bf16)dr; p[H + j] = (bf16)dz; p[2 * H + j] = (bf16)dnr;
                dn_s[(fq * 4 + jr) * LDH + j] = (bf16)dn;
            }
        }
        __syncthreads();
        // [t][row] operands of the batched weight gradients: dgh as it stands, dgi = dgh with the n-columns replaced by dn
        for (int v = tid; v < TR * (GP / 8); v += NTHR) {
            const int row = v / (GP / 8), c = v - row * (GP / 8);
            if (r0 + row < R) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(dg + row * LDGK + c * 8);
                const size_t o = ((size_t)t * R + r0 + row) * GP + c * 8;
                *reinterpret_cast<bf16x8*>(a.dgh_b + o) = x;
                const bool ncol = c * 8 >= 2 * H && c * 8 < 3 * H;
                *reinterpret_cast<bf16x8*>(a.dgi_b + o) = ncol ? *reinterpret_cast<const bf16x8*>(dn_s + row * LDH + c * 8 - 2 * H) : x;
            }
        }
        if (t > 0) {        // dh[t-1] += dgh[t] * W_hh  (A fragments in two halves: registers)
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                constexpr int HS = (KS + 1) / 2;
                bf16x8 af[HS];
#pragma unroll
                for (int q = 0; q < HS; ++q) {
                    const int s = half * HS + q;
                    if (s < KS) af[q] = *reinterpret_cast<const bf16x8*>(dg + fr * LDGK + s * 32 + fq * 8);
                }
#pragma unroll
                for (int q = 0; q < HS; ++q) {
                    const int s = half * HS + q;
                    if (s < KS) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q], w0[s], c0, 0, 0, 0);
                        if (two) c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                     af[q], *reinterpret_cast<const bf16x8*>(wl + ((size_t)(wave * KS + s) * 64 + lane) * 8), c1, 0, 0, 0);
                    }
                }
            }
            acc[0] = c0; acc[1] = c1;
        }
    }
}

// BPTT of the same recurrence: per step the gate backward and dh[t-1] += dgh[t] * W_hh (253 KB bf16 streamed).
__global__ __launch_bounds__(NTHR) void coco_enc_bwd_kernel(const CocoEncBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dhf = reinterpret_cast<float*>(smem);                    // [16][H] direct path d * z
    float* o = dhf + TR * H;                                        // [16][LDT] dgh[t+1] * W_hh
    bf16* dgh = reinterpret_cast<bf16*>(o + TR * LDT);              // [16][LDGK]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * TR, R = a.B, T = a.T;
    const size_t RH = (size_t)R * H;
    const int grow = tid >> 5, c0 = tid & 31;
    const bool gok = r0 + grow < R;
    const size_t gr = gok ? r0 + grow : 0;
    for (int i = tid; i < TR * H; i += NTHR) {
        const int row = i / H;
        dhf[i] = r0 + row < R ? a.dh_init[(size_t)(r0 + row) * H + (i - row * H)] : 0.f;
    }
    for (int i = tid; i < TR * LDT; i += NTHR) o[i] = 0.f;
    for (int i = tid; i < TR * LDGK; i += NTHR) dgh[i] = (bf16)0.f;
    constexpr int D = 2;                                            // hhT: 2 tile slots x 2 k-chunks = 4 chunks per step
    const WMat<GP / 32, HP / 16 - 1> d_hhT(a.w_hhT);
    bf16x8 ring[D][KCH];
#pragma unroll
    for (int q = 0; q < D; ++q) load_chunk(ring[q], d_hhT, q, wave, lane);
    float sr[NQH], sz[NQH], sn[NQH], sg[NQH], hp[NQH];
    auto fetch = [&](int t) {       // saved gates of step t and h[t-1]: requested one GEMM ahead of their use
        const float* s = a.sav + ((size_t)t * R + gr) * 4 * H;
        const float* hpp = a.h_all + (size_t)(t > 0 ? t - 1 : 0) * RH + gr * H;
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = min(c0 + 32 * q, H - 1);
            sr[q] = s[j]; sz[q] = s[H + j]; sn[q] = s[2 * H + j]; sg[q] = s[3 * H + j];
            hp[q] = t > 0 ? hpp[j] : 0.f;
        }
    };
    fetch(T - 1);
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
#pragma unroll
        for (int q = 0; q < NQH; ++q) {
            const int j = c0 + 32 * q;
            if (j < H) {
                const int i = grow * H + j;
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
                if (gok) {
                    const float r = sr[q], z = sz[q], n = sn[q], ghn = sg[q];
                    const float d = dhf[i] + o[grow * LDT + j];
                    dn = d * (1.0f - z) * (1.0f - n * n);
                    dz = d * (hp[q] - n) * z * (1.0f - z);
                    dr = dn * ghn * r * (1.0f - r);
                    dnr = dn * r;
                    dd = d * z;
                }
                dhf[i] = dd;
                dgh[grow * LDGK + j] = (bf16)dr; dgh[grow * LDGK + H + j] = (bf16)dz; dgh[grow * LDGK + 2 * H + j] = (bf16)dnr;
                if (gok) {
                    bf16* gi = a.dgi_b + ((size_t)t * R + gr) * GP;
                    bf16* gh = a.dgh_b + ((size_t)t * R + gr) * GP;
                    gi[j] = (bf16)dr; gi[H + j] = (bf16)dz; gi[2 * H + j] = (bf16)dn;
                    gh[j] = (bf16)dr; gh[H + j] = (bf16)dz; gh[2 * H + j] = (bf16)dnr;
                }
            }
        }
        __syncthreads();
        if (t > 0) {
            fetch(t - 1);
            stream_gemm<2, D, 0>(dgh, LDGK, d_hhT, o, LDT, ring, d_hhT, t > 1, wave, lane);
        }
        __syncthreads();
    }
}

// dst[(t*B + b)*ld + e] = bf16(src[(b*T + t)*E + e]) for e < E, 1.0 at e == E (the bias gradient rides in the weight gradient),
// 0 in the remaining pad columns: the rows are also the K-contiguous operand of the transposed input-projection GEMM, where a
// stale NaN in a pad column would poison the dot product even against a zero weight
__global__ __launch_bounds__(256) void text_tb_kernel(const float* src, int B, int T, int ld, bf16* dst) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * T * ld) return;
    const int e = (int)(i % ld);
    const long long bt = i / ld;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const float v = e < E ? src[bt * E + e] : (e == E ? 1.f : 0.f);
    dst[((size_t)t * B + b) * ld + e] = (bf16)v;
}

// out[r][c] = sum_t in[(t*R + r)*ld + c]
__global__ __launch_bounds__(256) void time_sum_bf16_kernel(const bf16* in, int T, int R, int ld, int cols, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const bf16* p = in + (size_t)r * ld + c;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += (float)p[(size_t)t * R * ld];
    out[i] = acc;
}
__global__ __launch_bounds__(256) void dout_combine_kernel(const float* __restrict__ dw, const float* __restrict__ fb, int T, int R, bf16* __restrict__ dout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * R * E) return;
    const long long m = i / E;                  // t * R + r
    const int e = (int)(i - m * E), t = (int)(m / R), r = (int)(m - (long long)t * R);
    float v = dw[((size_t)r * T + t) * E + e];
    if (t + 1 < T) v += fb[i];
    dout[m * EP + e] = (bf16)v;
}
__global__ __launch_bounds__(256) void dw16_kernel(const float* __restrict__ dw, long long rows, bf16* __restrict__ dw16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * XP) return;
    const long long r = i / XP;
    const int e = (int)(i - r * XP);
    dw16[i] = (bf16)(e < E ? dw[r * E + e] : 0.f);
}

}  // namespace

int launch_coco_time_sum_bf16(const bf16* in, int T, int R, int ld, int cols, float* out, hipStream_t s) {
    hipLaunchKernelGGL(time_sum_bf16_kernel, dim3((unsigned)(((long long)R * cols + 255) / 256)), dim3(256), 0, s, in, T, R, ld, cols, out);
    return mmvae_check_launch("coco_time_sum_bf16");
}
int launch_coco_dout_combine(const float* dw, const float* fb, int T, int R, bf16* dout, hipStream_t s) {
    hipLaunchKernelGGL(dout_combine_kernel, dim3((unsigned)(((long long)T * R * E + 255) / 256)), dim3(256), 0, s, dw, fb, T, R, dout);
    return mmvae_check_launch("coco_dout_combine");
}
int launch_coco_dw16(const float* dw, long long rows, bf16* dw16, hipStream_t s) {
    hipLaunchKernelGGL(dw16_kernel, dim3((unsigned)((rows * XP + 255) / 256)), dim3(256), 0, s, dw, rows, dw16);
    return mmvae_check_launch("coco_dw16");
}

int launch_coco_comb(const float* wih0, const float* bih0, const float* who, const float* bho, int D, const float* sos, bf16* comb, bf16* combT,
                     float* sosv, float* wz, float* bz, hipStream_t s) {
    MMVAE_REQUIRE(D >= 1 && H + D <= COMB_TPB, "coco_comb: n_latents = %d", D);
    hipLaunchKernelGGL(coco_comb_kernel, dim3(GP / 4), dim3(COMB_TPB), 0, s, wih0, bih0, who, bho, D, sos, comb, combT, sosv, wz, bz);
    return mmvae_check_launch("coco_comb");
}
int launch_coco_dec_fwd(const CocoDecFwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(2 * TR * LDG + 2 * TR * H + 3 * G) * sizeof(float) + (size_t)(TR * LDX + 3 * TR * LDH) * sizeof(bf16);
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) {
        auto big = [](auto kern) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
        big(&coco_dec_fwd_kernel<true, true>); big(&coco_dec_fwd_kernel<true, false>);
        big(&coco_dec_fwd_kernel<false, true>); big(&coco_dec_fwd_kernel<false, false>);
    }
    const bool save = a.sav0 != nullptr;
    if (save) MMVAE_REQUIRE(a.sav1 && a.h0_all && a.h1_all && a.xb_all && a.h0b_all && a.midb_all && a.h1b_all, "coco_dec_fwd: save buffers");
    if (a.cluster > 1) {        // P ranks per row block (coco_dec_fwd_cl_kernel); the caller zeroed cl_xchg / cl_timeout
        MMVAE_REQUIRE((a.cluster == 4 || a.cluster == 8) && a.wg_ih0 && a.wg_hh0 && a.wg_ih1 && a.wg_hh1 && a.cl_xchg && a.cl_timeout,
                      "coco_dec_fwd: cluster arguments");
        const int nblk = ceil_div(a.R, TR), nblk_pad = (nblk + 7) / 8 * 8;
        auto lds_of = [](int P) {
            const int nubmax = (13 + P - 1) / P;
            return (size_t)(2 * TR * (3 * nubmax * 16 + 4) + 3 * G) * sizeof(float) + (size_t)(TR * LDX + 3 * TR * LDH) * sizeof(bf16);
        };
        // every rank of a cluster must be resident for the exchange to complete: one workgroup per CU (LDS > 80 KB is not
        // needed for that: the grid never exceeds the CU count)
        MMVAE_REQUIRE(nblk_pad * a.cluster <= mmvae_cu_count(), "coco_dec_fwd: %d x %d workgroups do not fit the chip", nblk_pad, a.cluster);
        // (the dynamic-LDS limit is raised per kernel to what the launch asks for: a few microseconds of host time per step,
        //  and no silent dependence on the 64 KB default when the tile constants change)
        auto gc = [&](auto kern, int P) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(P));
            hipLaunchKernelGGL(kern, dim3(nblk_pad * P), dim3(NTHR), lds_of(P), s, a);
        };
        if (a.cluster == 8 && a.wg_comb) {      // composed form: two exchanges per step, weights resident
            MMVAE_REQUIRE(a.zi0p && a.sosv, "coco_dec_fwd: composed-form arguments");
            const size_t lds8 = (size_t)(2 * TR * (3 * 2 * 16 + 4) + TR * (3 * 16 + 4) + 3 * G) * sizeof(float) + (size_t)(3 * TR * LDH) * sizeof(bf16);
            auto g8 = [&](auto kern) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
                hipLaunchKernelGGL(kern, dim3(nblk_pad * 8), dim3(NTHR), lds8, s, a);
            };
            if (a.keep) { if (save) g8(&coco_dec_fwd_c8_kernel<true, true>); else g8(&coco_dec_fwd_c8_kernel<true, false>); }
            else { if (save) g8(&coco_dec_fwd_c8_kernel<false, true>); else g8(&coco_dec_fwd_c8_kernel<false, false>); }
            return mmvae_check_launch("coco_dec_fwd_c8");
        }
        if (a.cluster == 4) {
            if (a.keep) { if (save) gc(&coco_dec_fwd_cl_kernel<true, true, 4>, 4); else gc(&coco_dec_fwd_cl_kernel<true, false, 4>, 4); }
            else { if (save) gc(&coco_dec_fwd_cl_kernel<false, true, 4>, 4); else gc(&coco_dec_fwd_cl_kernel<false, false, 4>, 4); }
        } else {
            if (a.keep) { if (save) gc(&coco_dec_fwd_cl_kernel<true, true, 8>, 8); else gc(&coco_dec_fwd_cl_kernel<true, false, 8>, 8); }
            else { if (save) gc(&coco_dec_fwd_cl_kernel<false, true, 8>, 8); else gc(&coco_dec_fwd_cl_kernel<false, false, 8>, 8); }
        }
        return mmvae_check_launch("coco_dec_fwd_cl");
    }
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(ceil_div(a.R, TR)), dim3(NTHR), lds, s, a); };
    if (a.keep) { if (save) go(&coco_dec_fwd_kernel<true, true>); else go(&coco_dec_fwd_kernel<true, false>); }
    else { if (save) go(&coco_dec_fwd_kernel<false, true>); else go(&coco_dec_fwd_kernel<false, false>); }
    return mmvae_check_launch("coco_dec_fwd");
}
int launch_coco_dec_bwd(const CocoDecBwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(2 * TR * H + TR * LDO + 2 * TR * LDT) * sizeof(float) + (size_t)(TR * LDX + 2 * TR * LDGK) * sizeof(bf16);
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once))
    {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&coco_dec_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&coco_dec_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (a.cluster > 1) {        // P ranks per row block (coco_dec_bwd_cl_kernel); the caller zeroed cl_xchg / cl_timeout
        MMVAE_REQUIRE((a.cluster == 4 || a.cluster == 8) && a.cl_xchg && a.cl_timeout, "coco_dec_bwd: cluster arguments");
        const int nblk = ceil_div(a.R, TR), nblk_pad = (nblk + 7) / 8 * 8;
        MMVAE_REQUIRE(nblk_pad * a.cluster <= mmvae_cu_count(), "coco_dec_bwd: %d x %d workgroups do not fit the chip", nblk_pad, a.cluster);
        auto lds_of = [](int P) {
            const int nubmax = (13 + P - 1) / P, noemax = (19 + P - 1) / P, kh = NW / nubmax;
            return (size_t)(3 * kh * TR * (nubmax * 16 + 4) + kh * TR * (noemax * 16 + 4)) * sizeof(float) +
                   (size_t)(TR * LDX + 2 * TR * LDGK) * sizeof(bf16);
        };
        auto gc = [&](auto kern, int P) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(P));
            hipLaunchKernelGGL(kern, dim3(nblk_pad * P), dim3(NTHR), lds_of(P), s, a);
        };
        if (a.cluster == 8 && a.w_combT) {      // composed form: two exchanges per step, weights resident
            MMVAE_REQUIRE(a.dw16 && a.dzi0 && a.dzi1, "coco_dec_bwd: the composed form needs the bf16 copy of the loss gradient and the sum buffers");
            const size_t lds8 = (size_t)(4 * 4 * TR * (2 * 16 + 4)) * sizeof(float) + (size_t)(2 * TR * LDGK) * sizeof(bf16);
            static std::atomic<unsigned> once8{0};
            if (mmvae_first_use_on_device(once8)) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&coco_dec_bwd_c8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute(reinterpret_cast<const void*>(&coco_dec_bwd_c8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            if (a.keep) hipLaunchKernelGGL(coco_dec_bwd_c8_kernel<true>, dim3(nblk_pad * 8), dim3(NTHR), lds8, s, a);
            else hipLaunchKernelGGL(coco_dec_bwd_c8_kernel<false>, dim3(nblk_pad * 8), dim3(NTHR), lds8, s, a);
            return mmvae_check_launch("coco_dec_bwd_c8");
        }
        if (a.cluster == 4) { if (a.keep) gc(&coco_dec_bwd_cl_kernel<true, 4>, 4); else gc(&coco_dec_bwd_cl_kernel<false, 4>, 4); }
        else { if (a.keep) gc(&coco_dec_bwd_cl_kernel<true, 8>, 8); else gc(&coco_dec_bwd_cl_kernel<false, 8>, 8); }
        return mmvae_check_launch("coco_dec_bwd_cl");
    }
    if (a.keep) hipLaunchKernelGGL(coco_dec_bwd_kernel<true>, dim3(ceil_div(a.R, TR)), dim3(NTHR), lds, s, a);
    else hipLaunchKernelGGL(coco_dec_bwd_kernel<false>, dim3(ceil_div(a.R, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("coco_dec_bwd");
}

int launch_coco_enc_fwd(const CocoEncFwdArgs& a, hipStream_t s) {
    MMVAE_REQUIRE(a.h_all && (!a.sav || a.hb_all), "coco_enc_fwd: save buffers");
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) {
        auto big = [](auto kern) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
        big(&coco_enc_fwd_kernel<true>); big(&coco_enc_fwd_kernel<false>);
        big(&coco_enc_fwd_res_kernel<true>); big(&coco_enc_fwd_res_kernel<false>);
    }
    if (a.resident) {       // a.w_hh = three per-gate matrices, back to back; gi / sav / h_all with the batch row fastest
        MMVAE_REQUIRE(a.B % 4 == 0 && a.h_last && a.bih, "coco_enc_fwd (resident): batch must be a multiple of 4");
        const size_t lds = (size_t)(2 * TR * LDH + 5 * 3 * (HP / 32) * 64 * 8) * sizeof(bf16);
        if (a.sav) hipLaunchKernelGGL(coco_enc_fwd_res_kernel<true>, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
        else hipLaunchKernelGGL(coco_enc_fwd_res_kernel<false>, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
        return mmvae_check_launch("coco_enc_fwd_res");
    }
    const size_t lds = (size_t)(TR * LDG + TR * H + G) * sizeof(float) + (size_t)(TR * LDH) * sizeof(bf16);
    if (a.sav) hipLaunchKernelGGL(coco_enc_fwd_kernel<true>, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
    else hipLaunchKernelGGL(coco_enc_fwd_kernel<false>, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("coco_enc_fwd");
}
int launch_coco_enc_bwd(const CocoEncBwdArgs& a, hipStream_t s) {
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&coco_enc_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&coco_enc_bwd_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (a.resident) {
        MMVAE_REQUIRE(a.B % 4 == 0, "coco_enc_bwd (resident): batch must be a multiple of 4");
        const size_t lds = (size_t)(2 * TR * LDGK + 2 * TR * LDH + 5 * (GP / 32) * 64 * 8) * sizeof(bf16);
        hipLaunchKernelGGL(coco_enc_bwd_res_kernel, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
        return mmvae_check_launch("coco_enc_bwd_res");
    }
    const size_t lds = (size_t)(TR * H + TR * LDT) * sizeof(float) + (size_t)(TR * LDGK) * sizeof(bf16);
    hipLaunchKernelGGL(coco_enc_bwd_kernel, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("coco_enc_bwd");
}
int launch_coco_text_tb(const float* text, int B, int T, int ld, bf16* dst, hipStream_t s) {
    hipLaunchKernelGGL(text_tb_kernel, dim3((unsigned)(((long long)B * T * ld + 255) / 256)), dim3(256), 0, s, text, B, T, ld, dst);
    return mmvae_check_launch("coco_text_tb");
}
