// See gemm_f32.h.
#include "gemm_f32.h"

namespace {

// ------------------------------------------------------------------ generic fp32 GEMM  C[m][n] = sum_k A(m,k) B(k,n) (+ bias[n])

// one 32x32 output tile per workgroup: 2x2 tiles of v_mfma_f32_16x16x4_f32 (operand fragment: lane -> index lane%16,
// one k per lane group lane/16; accumulator: lane -> column lane%16, rows 4*(lane/16)..+3).  The KS waves of the
// workgroup split K (these GEMMs have 10..400 tiles: the K chain is the latency) and are summed through LDS.
// Within a 16-deep step, MFMA s of lane group q consumes k = k0 + 4q + s: an operand that is contiguous along k is
// then ONE 16-byte load per lane per step, a strided one four 4-byte loads of the same k's.
// bz / nz: this workgroup's chunk of a grid-level K split (0 / 1 when there is none)
template <int KS, bool AV, bool BV, int U>
__device__ __forceinline__ void gemm_f32_body(const F32Gemm& g, float* red, const int bz, const int nz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    if (m0 >= g.M || n0 >= g.N) return;            // pair launches: the grid covers the larger problem (uniform per workgroup)
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Both operands go through buffer descriptors: 32-bit byte offsets, and an invalid element (row or column past the
    // end, k past K, step past this wave's range) is an offset of 0xFFFFFFFF that the hardware range check turns into 0.
    // (With plain pointers the compiler predicates every guarded load behind a branch and waits for it before issuing
    // the next one: a strided operand then costs 16 serialized memory round trips per k-step -- measured 5x slower.)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned acs = (unsigned)g.a_cs * 4u, brs = (unsigned)g.b_rs * 4u;
    unsigned arow[2], brow[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m0 + t * 16 + fr, n = n0 + t * 16 + fr;
        aok[t] = m < g.M; bok[t] = n < g.N;
        arow[t] = (unsigned)m * (unsigned)g.a_rs * 4u;
        brow[t] = (unsigned)n * (unsigned)g.b_cs * 4u;
    }
    const int nst_all = (g.K + 15) / 16;
    const int zs0 = (int)((long long)nst_all * bz / nz), zs1 = (int)((long long)nst_all * (bz + 1) / nz);
    const int nst = zs1 - zs0;
    const int st0 = zs0 + nst * wave / KS, st1 = zs0 + nst * (wave + 1) / KS;
    // U steps per iteration: every load of the iteration is issued before its first MFMA (these GEMMs are a few steps
    // per wave, so the loop is a chain of memory latencies unless the loads overlap)
    for (int st = st0; st < st1; st += U) {
        float av[U][4][2], bv[U][4][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool sin = st + u < st1;
            const int kb = (st + u) * 16 + fq * 4;             // this lane group's 4 consecutive k
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (AV) {                                      // K % 4 == 0 (host): the 4 k are all inside or all outside
                    const bool ok = sin && aok[t] && kb < g.K;
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, ok ? arow[t] + (unsigned)kb * 4u : 0xFFFFFFFFu, 0, 0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) av[u][q][t] = v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = sin && aok[t] && kb + q < g.K;
                        av[u][q][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, ok ? arow[t] + (unsigned)(kb + q) * acs : 0xFFFFFFFFu, 0, 0));
                    }
                }
                if (BV) {
                    const bool ok = sin && bok[t] && kb < g.K;
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, ok ? brow[t] + (unsigned)kb * 4u : 0xFFFFFFFFu, 0, 0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[u][q][t] = v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = sin && bok[t] && kb + q < g.K;
                        bv[u][q][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, ok ? brow[t] + (unsigned)(kb + q) * brs : 0xFFFFFFFFu, 0, 0));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][q][i], bv[u][q][j], acc[i][j], 0, 0, 0);
    }
    if (KS > 1) {
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((wave - 1) * 16 + (i * 2 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < KS - 1; ++w)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += red[(w * 16 + (i * 2 + j) * 4 + r) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + j * 16 + fr;
            if (n >= g.N) continue;
            const bool first = bz == 0;
            const float bias = (g.bias && first) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + fq * 4 + r;
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bias;
                if (g.addm && first) v += g.addm[(size_t)m * g.ldadd + n];
                float* c = g.C + (size_t)m * g.ldc + n;
                if (nz > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
}
template <int KS, bool AV, bool BV, int U>
__global__ __launch_bounds__(KS * 64) void gemm_f32_kernel(const F32Gemm g) {
    __shared__ float red[KS > 1 ? (KS - 1) * 16 * 64 : 1];
    gemm_f32_body<KS, AV, BV, U>(g, red, blockIdx.z, gridDim.z);
}
// two independent GEMMs in one launch: blockIdx.z picks the problem (no grid-level K split)
struct F32GemmPair { F32Gemm p[2]; };
template <int KS, bool AV, bool BV, int U>
__global__ __launch_bounds__(KS * 64) void gemm_f32_pair_kernel(const F32GemmPair pr) {
    __shared__ float red[KS > 1 ? (KS - 1) * 16 * 64 : 1];
    if (blockIdx.z == 0) gemm_f32_body<KS, AV, BV, U>(pr.p[0], red, 0, 1);
    else gemm_f32_body<KS, AV, BV, U>(pr.p[1], red, 0, 1);
}
template <int KS, int U>
int gemm_f32_pair_ks(const F32GemmPair& pr, bool av, bool bv, hipStream_t s) {
    const F32Gemm &a = pr.p[0], &b = pr.p[1];
    dim3 grid(ceil_div(a.M > b.M ? a.M : b.M, 32), ceil_div(a.N > b.N ? a.N : b.N, 32), 2), block(KS * 64);
    if (av && bv) hipLaunchKernelGGL((gemm_f32_pair_kernel<KS, true, true, U>), grid, block, 0, s, pr);
    else if (av) hipLaunchKernelGGL((gemm_f32_pair_kernel<KS, true, false, U>), grid, block, 0, s, pr);
    else if (bv) hipLaunchKernelGGL((gemm_f32_pair_kernel<KS, false, true, U>), grid, block, 0, s, pr);
    else hipLaunchKernelGGL((gemm_f32_pair_kernel<KS, false, false, U>), grid, block, 0, s, pr);
    return mmvae_check_launch("gemm_f32_pair");
}
template <int KS, int U>
int gemm_f32_ks(const F32Gemm& g, bool av, bool bv, hipStream_t s) {
    dim3 grid(ceil_div(g.M, 32), ceil_div(g.N, 32), g.ksplit > 1 ? g.ksplit : 1), block(KS * 64);
    if (av && bv) hipLaunchKernelGGL((gemm_f32_kernel<KS, true, true, U>), grid, block, 0, s, g);
    else if (av) hipLaunchKernelGGL((gemm_f32_kernel<KS, true, false, U>), grid, block, 0, s, g);
    else if (bv) hipLaunchKernelGGL((gemm_f32_kernel<KS, false, true, U>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<KS, false, false, U>), grid, block, 0, s, g);
    return mmvae_check_launch("gemm_f32");
}
}  // namespace

static int check_f32(const F32Gemm& g, bool& av, bool& bv) {
    MMVAE_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.A && g.B && g.C, "gemm_f32: bad arguments");
    MMVAE_REQUIRE(g.ksplit <= 1 || g.ksplit <= (g.K + 15) / 16, "gemm_f32: ksplit=%d for K=%d", g.ksplit, g.K);
    // operands are addressed with 32-bit byte offsets from their base pointers
    const long long amax = ((long long)(g.M - 1) * g.a_rs + (long long)(g.K - 1) * g.a_cs + 4) * 4;
    const long long bmax = ((long long)(g.N - 1) * g.b_cs + (long long)(g.K - 1) * g.b_rs + 4) * 4;
    MMVAE_REQUIRE(g.a_rs >= 0 && g.a_cs >= 0 && g.b_rs >= 0 && g.b_cs >= 0 && amax < 0x7FFFFFFFll && bmax < 0x7FFFFFFFll,
                  "gemm_f32: operand spans more than 2 GiB (or has a negative stride)");
    // 16-byte loads along k need a unit k stride, K a multiple of 4, rows that start 16-byte aligned
    av = g.a_cs == 1 && g.K % 4 == 0 && g.a_rs % 4 == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0;
    bv = g.b_rs == 1 && g.K % 4 == 0 && g.b_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0;
    return MMVAE_OK;
}

int gemm_f32_pair(const F32Gemm& a, const F32Gemm& b, hipStream_t s) {
    bool av0, bv0, av1, bv1;
    MMVAE_TRY(check_f32(a, av0, bv0));
    MMVAE_TRY(check_f32(b, av1, bv1));
    MMVAE_REQUIRE(a.ksplit <= 1 && b.ksplit <= 1, "gemm_f32_pair: no grid-level K split");
    F32GemmPair pr; pr.p[0] = a; pr.p[1] = b;
    const bool av = av0 && av1, bv = bv0 && bv1;
    const int tiles = ceil_div(a.M, 32) * ceil_div(a.N, 32) + ceil_div(b.M, 32) * ceil_div(b.N, 32);
    const int per = ceil_div(a.K > b.K ? a.K : b.K, 16);
    if (tiles <= 128 && per >= 16) return per <= 48 ? gemm_f32_pair_ks<8, 6>(pr, av, bv, s) : gemm_f32_pair_ks<8, 4>(pr, av, bv, s);
    if (tiles <= 256 && per >= 8) return per <= 24 ? gemm_f32_pair_ks<4, 6>(pr, av, bv, s) : gemm_f32_pair_ks<4, 4>(pr, av, bv, s);
    if (tiles <= 512 && per >= 4) return gemm_f32_pair_ks<2, 4>(pr, av, bv, s);
    return gemm_f32_pair_ks<1, 4>(pr, av, bv, s);
}

int gemm_f32(const F32Gemm& g, hipStream_t s) {
    bool av, bv;
    MMVAE_TRY(check_f32(g, av, bv));
    const int tiles = ceil_div(g.M, 32) * ceil_div(g.N, 32), nst = ceil_div(g.K, 16);
    const int nz = g.ksplit > 1 ? g.ksplit : 1;
    // few tiles: the K chain is the latency -- split it over the waves of the workgroup, and when a wave's share fits one
    // batch of U = 6 steps, issue all of its loads at once
    const int per = nst / nz;
    if (tiles * nz <= 128 && per >= 16) return per <= 48 ? gemm_f32_ks<8, 6>(g, av, bv, s) : gemm_f32_ks<8, 4>(g, av, bv, s);
    if (tiles * nz <= 256 && per >= 8) return per <= 24 ? gemm_f32_ks<4, 6>(g, av, bv, s) : gemm_f32_ks<4, 4>(g, av, bv, s);
    if (tiles * nz <= 512 && per >= 4) return gemm_f32_ks<2, 4>(g, av, bv, s);
    return gemm_f32_ks<1, 4>(g, av, bv, s);
}
