// Persistent row-tile kernels of the MultiMNIST text encoder / decoder (fwd + bwd).
// One workgroup = 4 waves = 16 batch rows for the whole recurrence.  Matmuls: 16 x N x K MFMA tiles
// (v_mfma_f32_16x16x32_bf16), A fragments from LDS (row stride K+8 elements: conflict-free ds_read_b128),
// B fragments straight from the packed bf16 weights in L2 (each 16-lane group reads 64 contiguous bytes per row).
// Gate math, softmax, NLL and all state are fp32.
#include "text.h"
#include <math.h>

namespace {

constexpr int TR = 16;       // rows per workgroup
constexpr int NW = 8;        // waves per workgroup (2 per SIMD: one streams weights while the other does gate math)
constexpr int NTHR = NW * 64;
constexpr int H = TXT_H;
constexpr int HP = TXT_HP;
constexpr int GL = TXT_G3P;  // fp32 gate buffer row stride (304)
constexpr int GK = TXT_G3K;  // 320

__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }

// out[16][ldo] = A[16][32*ksteps] * Wp[16*ntiles][kpad]^T   (Wp fragment-major)
// Each wave owns the column tiles wave, wave+NW, ... (at most MAXT).  ALL of its weight fragments (MAXT x ksteps
// independent 16-byte loads, clamped so that no branch surrounds a load) are issued before the first MFMA: one L2
// round trip per GEMM instead of one per tile -- the recurrence is a latency chain, not a bandwidth problem.
// KS = compile-time bound on the k-steps, MAXT = tiles per wave (ceil(ntiles / NW)).
template <int KS, int MAXT>
__device__ __forceinline__ void rowtile_gemm(const bf16* A, int lda, int ksteps, const bf16* __restrict__ Wp, int kpad, int ntiles,
                                             float* out, int ldo, int wave, int lane) {
    constexpr int MAXKS = KS;
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 bw[MAXT][MAXKS];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int nt = min(wave + t * NW, ntiles - 1);
        // fragment-major packs (PackDesc::frag): the 64 lanes' vectors of one (tile, k-step) are one contiguous 1 KB block, so
        // a wave's load is fully coalesced (row-major rows touched 16 half-used cache lines per instruction)
        const bf16* w = Wp + ((size_t)nt * (kpad >> 5) * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) bw[t][ks] = *reinterpret_cast<const bf16x8*>(w + min(ks, ksteps - 1) * 512);
    }
    bf16x8 af[MAXKS];
#pragma unroll
    for (int ks = 0; ks < MAXKS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(A + fr * lda + min(ks, ksteps - 1) * 32 + fq * 8);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int nt = wave + t * NW;
        if (nt >= ntiles) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < ksteps) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bw[t][ks], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(fq * 4 + j) * ldo + nt * 16 + fr] = acc[j];
    }
}

// The same product with the wave's weight fragments held by the CALLER (registers that live across the whole recurrence, or an LDS
// copy of the fragment-major pack): the weights of a GRU do not change between time steps, and re-reading them from L2 was 6-8 us
// of every GEMM phase (in-kernel stamps: 264 KB per workgroup and step at about half the per-CU L2 rate).
template <int KS, int MAXT>
__device__ __forceinline__ void load_wfrags(bf16x8 (&bw)[MAXT][KS], const bf16* __restrict__ Wp, int ntiles, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int nt = min(wave + t * NW, ntiles - 1);
        const bf16* w = Wp + ((size_t)nt * KS * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bw[t][ks] = *reinterpret_cast<const bf16x8*>(w + ks * 512);
    }
}
template <int KS, int MAXT>
__device__ __forceinline__ void rowtile_gemm_w(const bf16* A, int lda, const bf16x8 (&bw)[MAXT][KS], int ntiles, float* out, int ldo, int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 af[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(A + fr * lda + ks * 32 + fq * 8);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int nt = wave + t * NW;
        if (nt >= ntiles) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bw[t][ks], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(fq * 4 + j) * ldo + nt * 16 + fr] = acc[j];
    }
}
// ... with the fragments of k-steps 0..2 in LDS (wl: [tile][3][64 lanes][8], one conflict-free ds_read_b128 per fragment) and k-step 3 in registers
__device__ __forceinline__ void rowtile_gemm_lds3(const bf16* A, int lda, const bf16* wl, const bf16x8 (&wk3)[3], int ntiles, float* out, int ldo, int wave, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 af[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(A + fr * lda + ks * 32 + fq * 8);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nt = wave + t * NW;
        if (nt >= ntiles) break;
        const bf16* w = wl + ((size_t)nt * 3 * 64 + lane) * 8;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], *reinterpret_cast<const bf16x8*>(w + ks * 512), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[3], wk3[t], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(fq * 4 + j) * ldo + nt * 16 + fr] = acc[j];
    }
}

// GRU gate math for 16 rows (PyTorch gate order r,z,n). hf: fp32 state in/out, hb: bf16 copy for the next MFMA.
// save: [5][R][100] (r, z, n, W_hn h + b_hn, h_prev) or null.
__device__ __forceinline__ void gru_gates(const float* gi, const float* gh, const float* bih,
                                          const float* bhh, float* hf, bf16* hb, int ldh, int r0, int R,
                                          float* save, int tid) {
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        const int row = idx / H, j = idx - row * H;
        const float* a = gi + row * GL;
        const float* b = gh + row * GL;
        float r = sigm(a[j] + bih[j] + b[j] + bhh[j]);
        float z = sigm(a[H + j] + bih[H + j] + b[H + j] + bhh[H + j]);
        float ghn = b[2 * H + j] + bhh[2 * H + j];
        float n = tanh_fast(a[2 * H + j] + bih[2 * H + j] + r * ghn);
        float hp = hf[row * H + j];
        float hn = (1.0f - z) * n + z * hp;
        hf[row * H + j] = hn;
        hb[row * ldh + j] = (bf16)hn;
        if (save && r0 + row < R) {
            const size_t o = (size_t)(r0 + row) * H + j, pl = (size_t)R * H;
            save[o] = r; save[pl + o] = z; save[2 * pl + o] = n; save[3 * pl + o] = ghn; save[4 * pl + o] = hp;
        }
    }
}

// backward of gru_gates: dh (fp32 [16][100], grad wrt h_new) -> dgi/dgh (bf16 LDS [16][ldg] + global), dh_direct = dh*z
__device__ __forceinline__ void gru_gates_bwd(const float* save, int r0, int R, const float* dh, bf16* dgi_s, bf16* dgh_s, int ldg,
                                              float* dh_direct, bf16* dgi_g, bf16* dgh_g, int tid) {
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        const int row = idx / H, j = idx - row * H;
        float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dd = 0.f;
        if (r0 + row < R) {
            const size_t o = (size_t)(r0 + row) * H + j, pl = (size_t)R * H;
            const float r = save[o], z = save[pl + o], n = save[2 * pl + o], ghn = save[3 * pl + o], hp = save[4 * pl + o];
            const float d = dh[row * H + j];
            dn = d * (1.0f - z) * (1.0f - n * n);
            dz = d * (hp - n) * z * (1.0f - z);
            dr = dn * ghn * r * (1.0f - r);
            dnr = dn * r;
            dd = d * z;
        }
        dh_direct[row * H + j] = dd;
        dgi_s[row * ldg + j] = (bf16)dr; dgi_s[row * ldg + H + j] = (bf16)dz; dgi_s[row * ldg + 2 * H + j] = (bf16)dn;
        dgh_s[row * ldg + j] = (bf16)dr; dgh_s[row * ldg + H + j] = (bf16)dz; dgh_s[row * ldg + 2 * H + j] = (bf16)dnr;
        if (r0 + row < R) {
            const size_t g = (size_t)(r0 + row) * GL;
            if (dgi_g) { dgi_g[g + j] = (bf16)dr; dgi_g[g + H + j] = (bf16)dz; dgi_g[g + 2 * H + j] = (bf16)dn; }
            if (dgh_g) { dgh_g[g + j] = (bf16)dr; dgh_g[g + H + j] = (bf16)dz; dgh_g[g + 2 * H + j] = (bf16)dnr; }
        }
    }
}

// per-thread column sums of a bf16 LDS gate-gradient tile (thread j < 300 owns column j)
__device__ __forceinline__ float colsum16(const bf16* t, int ld, int col) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < TR; ++r) s += (float)t[r * ld + col];
    return s;
}

__device__ __forceinline__ void zero_bf(bf16* p, int n, int tid) {
    for (int i = tid; i < n; i += NTHR) p[i] = (bf16)0.f;
}
__device__ __forceinline__ void zero_f(float* p, int n, int tid) {
    for (int i = tid; i < n; i += NTHR) p[i] = 0.f;
}
// copy LDS bf16 tile rows [16][ld] (first `w` columns) to global [R][gld]
__device__ __forceinline__ void save_tile(const bf16* t, int ld, int w, bf16* g, int gld, int r0, int R, int tid) {
    if (!g) return;
    for (int idx = tid; idx < TR * w; idx += NTHR) {
        int row = idx / w, c = idx - row * w;
        if (r0 + row < R) g[(size_t)(r0 + row) * gld + c] = t[row * ld + c];
    }
}

// ============================================================== text encoder forward
__global__ __launch_bounds__(NTHR) void text_encoder_fwd_kernel(const TextEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LH = HP + 8;
    float* gi = reinterpret_cast<float*>(smem);              // [16][304]
    float* gh = gi + TR * GL;                                // [16][304]
    float* hf = gh + TR * GL;                                // [16][100]
    float* hr = hf + TR * H;                                 // [16][100] reverse-direction state
    bf16* xb = reinterpret_cast<bf16*>(hr + TR * H);         // [16][136]
    bf16* hb = xb + TR * LH;                                 // [16][136]
    bf16* hrb = hb + TR * LH;                                // [16][136]
    float* cst = reinterpret_cast<float*>(hrb + TR * LH);   // embed[1200] | bih_f bhh_f bih_r bhh_r [1200]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * TR, R = a.B;
    zero_bf(xb, 3 * TR * LH, tid);
    zero_f(hf, 2 * TR * H, tid);
    float* c_emb = cst; float* c_b = cst + 1200;
    for (int i = tid; i < 1200; i += NTHR) c_emb[i] = a.embed[i];
    for (int i = tid; i < 300; i += NTHR) {
        c_b[i] = a.fwd.bih[i]; c_b[300 + i] = a.fwd.bhh[i]; c_b[600 + i] = a.rev.bih[i]; c_b[900 + i] = a.rev.bhh[i];
    }
    __syncthreads();
    const size_t plane = (size_t)R * HP;
    for (int t = 0; t < TXT_T; ++t) {
        for (int idx = tid; idx < TR * H; idx += NTHR) {
            int row = idx / H, j = idx - row * H;
            int tok = (r0 + row < R) ? (int)a.tokens[(size_t)(r0 + row) * TXT_T + t] : 0;
            xb[row * LH + j] = (bf16)c_emb[tok * H + j];
        }
        __syncthreads();
        save_tile(xb, LH, HP, a.x_bf ? a.x_bf + t * plane : nullptr, HP, r0, R, tid);
        save_tile(hb, LH, HP, a.hprev_bf ? a.hprev_bf + t * plane : nullptr, HP, r0, R, tid);
        rowtile_gemm<4, 3>(xb, LH, HP / 32, a.fwd.wih, a.fwd.kih, GL / 16, gi, GL, wave, lane);
        rowtile_gemm<4, 3>(hb, LH, HP / 32, a.fwd.whh, HP, GL / 16, gh, GL, wave, lane);
        __syncthreads();
        gru_gates(gi, gh, c_b, c_b + 300, hf, hb, LH, r0, R, a.gates_f ? a.gates_f + (size_t)t * 5 * R * H : nullptr, tid);
        __syncthreads();
    }
    // reverse direction at the last time step: one step from h = 0 on token T-1 (xb still holds it)
    rowtile_gemm<4, 3>(xb, LH, HP / 32, a.rev.wih, a.rev.kih, GL / 16, gi, GL, wave, lane);
    zero_f(gh, TR * GL, tid);       // W_hh * 0
    __syncthreads();
    gru_gates(gi, gh, c_b + 600, c_b + 900, hr, hrb, LH, r0, R, a.gates_r, tid);
    __syncthreads();
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        int row = idx / H, j = idx - row * H;
        xb[row * LH + j] = (bf16)(hf[row * H + j] + hr[row * H + j]);    // x[:, :H] + x[:, H:]  (model.py:245)
    }
    __syncthreads();
    save_tile(xb, LH, HP, a.hsum_bf, HP, r0, R, tid);
    const int D2 = 2 * a.D;
    rowtile_gemm<4, 2>(xb, LH, HP / 32, a.h2p, HP, a.nh2p / 16, gi, GL, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < TR * D2; idx += NTHR) {
        int row = idx / D2, j = idx - row * D2;
        if (r0 + row < R) a.out[(size_t)(r0 + row) * D2 + j] = gi[row * GL + j] + a.h2p_bias[j];
    }
}

// ============================================================== text encoder backward
__global__ __launch_bounds__(NTHR) void text_encoder_bwd_kernel(const TextEncBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TextEncArgs& f = a.f;
    constexpr int LG = GK + 8;                                // 328
    const int D2 = 2 * f.D, K2 = round_up(D2, 32), LD2 = K2 + 8;
    float* o1 = reinterpret_cast<float*>(smem);              // [16][128] gemm output
    float* dh = o1 + TR * HP;                                // [16][100]
    float* dhd = dh + TR * H;                                // [16][100] direct term dh*z
    float* demb = dhd + TR * H;                              // [12][100]
    bf16* dgi = reinterpret_cast<bf16*>(demb + TXT_V * H);   // [16][328]
    bf16* dgh = dgi + TR * LG;                               // [16][328]
    bf16* dob = dgh + TR * LG;                               // [16][LD2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * TR, R = f.B;
    zero_bf(dgi, 2 * TR * LG + TR * LD2, tid);
    zero_f(demb, TXT_V * H, tid);
    __syncthreads();
    float bsum = 0.f;      // h2p bias gradient (thread j < 2D)
    for (int idx = tid; idx < TR * D2; idx += NTHR) {
        int row = idx / D2, j = idx - row * D2;
        float v = (r0 + row < R) ? a.d_out[(size_t)(r0 + row) * D2 + j] : 0.f;
        dob[row * LD2 + j] = (bf16)v;
        if (a.d_out_bf && r0 + row < R) a.d_out_bf[(size_t)(r0 + row) * round_up(D2, 8) + j] = (bf16)v;
    }
    __syncthreads();
    if (tid < D2) {
        for (int r = 0; r < TR; ++r) bsum += (float)dob[r * LD2 + tid];
        atomicAdd(a.g_h2p_bias + tid, bsum);
    }
    // d(hf + hr) = d_out * W_h2p
    rowtile_gemm<8, 1>(dob, LD2, K2 / 32, f.h2pT, K2, 7, o1, HP, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < TR * H; idx += NTHR) dh[idx] = o1[(idx / H) * HP + idx % H];
    __syncthreads();
    // ---- reverse direction (single step from h=0 on token T-1)
    float gb[4][2] = {};                  // bih_f, bhh_f, bih_r, bhh_r partial sums (thread owns columns tid, tid+256)
    gru_gates_bwd(f.gates_r, r0, R, dh, dgi, dgh, LG, dhd, a.dgi_r, nullptr, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tid + q * NTHR < TXT_G3) { gb[2][q] += colsum16(dgi, LG, tid + q * NTHR); gb[3][q] += colsum16(dgh, LG, tid + q * NTHR); }
    rowtile_gemm<10, 1>(dgi, LG, GK / 32, f.rev.wihT, GK, 7, o1, HP, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        int row = idx / H, j = idx - row * H;
        if (r0 + row < R) atomicAdd(demb + (int)f.tokens[(size_t)(r0 + row) * TXT_T + TXT_T - 1] * H + j, o1[row * HP + j]);
    }
    __syncthreads();
    // ---- forward direction BPTT
    for (int t = TXT_T - 1; t >= 0; --t) {
        gru_gates_bwd(f.gates_f + (size_t)t * 5 * R * H, r0, R, dh, dgi, dgh, LG, dhd,
                      a.dgi_f + (size_t)t * R * GL, a.dgh_f + (size_t)t * R * GL, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + q * NTHR < TXT_G3) { gb[0][q] += colsum16(dgi, LG, tid + q * NTHR); gb[1][q] += colsum16(dgh, LG, tid + q * NTHR); }
        rowtile_gemm<10, 1>(dgi, LG, GK / 32, f.fwd.wihT, GK, 7, o1, HP, wave, lane);
        __syncthreads();
        for (int idx = tid; idx < TR * H; idx += NTHR) {
            int row = idx / H, j = idx - row * H;
            if (r0 + row < R) atomicAdd(demb + (int)f.tokens[(size_t)(r0 + row) * TXT_T + t] * H + j, o1[row * HP + j]);
        }
        __syncthreads();
        rowtile_gemm<10, 1>(dgh, LG, GK / 32, f.fwd.whhT, GK, 7, o1, HP, wave, lane);
        __syncthreads();
        for (int idx = tid; idx < TR * H; idx += NTHR) dh[idx] = dhd[idx] + o1[(idx / H) * HP + idx % H];
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tid + q * NTHR < TXT_G3) {
            const int c = tid + q * NTHR;
            atomicAdd(a.g_bih_f + c, gb[0][q]); atomicAdd(a.g_bhh_f + c, gb[1][q]);
            atomicAdd(a.g_bih_r + c, gb[2][q]); atomicAdd(a.g_bhh_r + c, gb[3][q]);
        }
    for (int idx = tid; idx < TXT_V * H; idx += NTHR)
        if (demb[idx] != 0.f) atomicAdd(a.g_embed + idx, demb[idx]);
}

// ============================================================== text decoder forward
struct DecLds {
    float *gi, *gh, *h0f, *h1f, *zf, *lg, *cst;     // cst: embed[12*100] | bih0 bhh0 bih1 bhh1 [4*300] | h2o_bias[16] | z2h_bias[100]
    bf16 *x0b, *zb, *h0b, *midb, *h1b, *hzb;
    int LX, LZ;
};
__device__ __forceinline__ DecLds dec_lds(char* smem, int kx, int kz) {
    DecLds L;
    L.LX = kx + 8; L.LZ = kz + 8;
    L.gi = reinterpret_cast<float*>(smem);
    L.gh = L.gi + TR * GL;
    L.h0f = L.gh + TR * GL;
    L.h1f = L.h0f + TR * H;
    L.zf = L.h1f + TR * H;               // [16][128]
    L.lg = L.zf + TR * 128;              // [16][16]
    L.cst = L.lg + TR * 16;              // [2528]
    L.x0b = reinterpret_cast<bf16*>(L.cst + 2528);
    L.hzb = L.x0b + TR * L.LX;
    L.zb = L.hzb + TR * L.LX;
    L.h0b = L.zb + TR * L.LZ;
    L.midb = L.h0b + TR * (HP + 8);
    L.h1b = L.midb + TR * (HP + 8);
    return L;
}
__host__ __device__ inline size_t dec_lds_bytes(int kx, int kz) {
    return (size_t)(2 * TR * GL + 2 * TR * H + TR * 128 + TR * 16 + 2528) * 4 +
           (size_t)(2 * TR * (kx + 8) + TR * (kz + 8) + 3 * TR * (HP + 8)) * 2;
}

__global__ __launch_bounds__(NTHR) void text_decoder_fwd_kernel(const TextDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int ctok[TR];
    DecLds L = dec_lds(smem, a.kx, a.kz);
    constexpr int LH = HP + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * TR, R = a.R, D = a.D;
    zero_bf(L.x0b, 2 * TR * L.LX + TR * L.LZ + 3 * TR * LH, tid);
    if (tid < TR) ctok[tid] = 10;     // SOS (multimnist/utils.py:17)
    // small read-only vectors live in LDS for the whole recurrence (no global round trip inside a step)
    float* c_emb = L.cst; float* c_b = L.cst + 1200; float* c_h2o = L.cst + 2400; float* c_z2h = L.cst + 2416;
    for (int i = tid; i < 1200; i += NTHR) c_emb[i] = a.embed[i];
    for (int i = tid; i < 300; i += NTHR) {
        c_b[i] = a.l0.bih[i]; c_b[300 + i] = a.l0.bhh[i]; c_b[600 + i] = a.l1.bih[i]; c_b[900 + i] = a.l1.bhh[i];
    }
    if (tid < TXT_V) c_h2o[tid] = a.h2o_bias[tid];
    if (tid < H) c_z2h[tid] = a.z2h_bias[tid];
    __syncthreads();
    for (int idx = tid; idx < TR * D; idx += NTHR) {
        int row = idx / D, j = idx - row * D;
        float z = (r0 + row < R) ? a.z[(size_t)(r0 + row) * D + j] : 0.f;
        L.zf[row * 128 + j] = z;
        bf16 zb = (bf16)z;
        L.zb[row * L.LZ + j] = zb;
        L.x0b[row * L.LX + H + j] = zb;
        L.hzb[row * L.LX + H + j] = zb;
    }
    __syncthreads();
    save_tile(L.zb, L.LZ, a.kz, a.z_bf, a.kz, r0, R, tid);
    // h = z2h(z) replicated into both layers (model.py:280)
    rowtile_gemm<4, 1>(L.zb, L.LZ, a.kz / 32, a.z2h, a.kz, 7, L.gi, GL, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        int row = idx / H, j = idx - row * H;
        float h = L.gi[row * GL + j] + c_z2h[j];
        L.h0f[idx] = h; L.h1f[idx] = h;
        L.h0b[row * LH + j] = (bf16)h; L.h1b[row * LH + j] = (bf16)h;
    }
    __syncthreads();
    float nll[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t pl128 = (size_t)R * HP, plx = (size_t)R * a.kx;
    for (int i = 0; i < TXT_T; ++i) {
        // c_in = swish(embed(c_in)) ; x0 = [c_in | z]   (model.py:299-300)
        for (int idx = tid; idx < TR * H; idx += NTHR) {
            int row = idx / H, j = idx - row * H;
            float e = c_emb[ctok[row] * H + j];
            L.x0b[row * L.LX + j] = (bf16)(e / (1.0f + expf(-e)));
        }
        __syncthreads();
        save_tile(L.x0b, L.LX, a.kx, a.x0_bf ? a.x0_bf + i * plx : nullptr, a.kx, r0, R, tid);
        save_tile(L.h0b, LH, HP, a.h0p_bf ? a.h0p_bf + i * pl128 : nullptr, HP, r0, R, tid);
        save_tile(L.h1b, LH, HP, a.h1p_bf ? a.h1p_bf + i * pl128 : nullptr, HP, r0, R, tid);
        rowtile_gemm<8, 3>(L.x0b, L.LX, a.kx / 32, a.l0.wih, a.l0.kih, GL / 16, L.gi, GL, wave, lane);
        rowtile_gemm<4, 3>(L.h0b, LH, HP / 32, a.l0.whh, HP, GL / 16, L.gh, GL, wave, lane);
        __syncthreads();
        gru_gates(L.gi, L.gh, c_b, c_b + 300, L.h0f, L.h0b, LH, r0, R,
                  a.gates ? a.gates + (size_t)(i * 2 + 0) * 5 * R * H : nullptr, tid);
        __syncthreads();
        // inter-layer dropout (nn.GRU dropout=0.1, train only)
        for (int idx = tid; idx < TR * H; idx += NTHR) {
            int row = idx / H, j = idx - row * H;
            float v = L.h0f[idx];
            if (a.keep && r0 + row < R) v = a.keep[((size_t)i * R + r0 + row) * H + j] ? v * a.keep_scale : 0.f;
            L.midb[row * LH + j] = (bf16)v;
        }
        __syncthreads();
        save_tile(L.midb, LH, HP, a.mid_bf ? a.mid_bf + i * pl128 : nullptr, HP, r0, R, tid);
        rowtile_gemm<4, 3>(L.midb, LH, HP / 32, a.l1.wih, a.l1.kih, GL / 16, L.gi, GL, wave, lane);
        rowtile_gemm<4, 3>(L.h1b, LH, HP / 32, a.l1.whh, HP, GL / 16, L.gh, GL, wave, lane);
        __syncthreads();
        gru_gates(L.gi, L.gh, c_b + 600, c_b + 900, L.h1f, L.h1b, LH, r0, R,
                  a.gates ? a.gates + (size_t)(i * 2 + 1) * 5 * R * H : nullptr, tid);
        __syncthreads();
        for (int idx = tid; idx < TR * H; idx += NTHR) {
            int row = idx / H, j = idx - row * H;
            L.hzb[row * L.LX + j] = L.h1b[row * LH + j];
        }
        __syncthreads();
        save_tile(L.hzb, L.LX, a.kx, a.hz_bf ? a.hz_bf + i * plx : nullptr, a.kx, r0, R, tid);
        rowtile_gemm<8, 1>(L.hzb, L.LX, a.kx / 32, a.h2o, a.kx, 1, L.lg, 16, wave, lane);
        __syncthreads();
        if (tid < TR) {
            const int row = tid;
            float v[TXT_V], mx = -INFINITY;
            for (int c = 0; c < TXT_V; ++c) { v[c] = L.lg[row * 16 + c] + c_h2o[c]; mx = fmaxf(mx, v[c]); }
            float se = 0.f;
            for (int c = 0; c < TXT_V; ++c) se += expf(v[c] - mx);
            const float lse = mx + logf(se);
            int best = 0; float bv = -INFINITY;
            for (int c = 0; c < TXT_V; ++c) { v[c] -= lse; if (v[c] > bv) { bv = v[c]; best = c; } }   // first max, like torch.max
            if (r0 + row < R) {
                const size_t o = ((size_t)(r0 + row) * TXT_T + i) * TXT_V;
                for (int c = 0; c < TXT_V; ++c) a.words[o + c] = v[c];
                if (a.tokens_out) a.tokens_out[(size_t)(r0 + row) * TXT_T + i] = best;
                if (a.target) {
                    const int pass = (r0 + row) / a.rows_per_pass, b = (r0 + row) - pass * a.rows_per_pass;
                    const int tg = (int)a.target[(size_t)b * TXT_T + i];
                    float vt = 0.f;
#pragma unroll
                    for (int c = 0; c < TXT_V; ++c) vt = (c == tg) ? v[c] : vt;
#pragma unroll
                    for (int p = 0; p < 4; ++p) nll[p] += (p == (pass & 3)) ? -vt : 0.f;
                    if (a.dwords)
                        for (int c = 0; c < TXT_V; ++c) a.dwords[o + c] = (c == tg) ? -a.nll_coef[pass & 3] : 0.f;
                }
                ctok[row] = a.force_tokens ? (int)a.force_tokens[(size_t)(r0 + row) * TXT_T + i] : best;
            }
        }
        __syncthreads();
    }
    if (a.target && a.nll_sum && tid < TR) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (nll[p] != 0.f) atomicAdd(a.nll_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + p, nll[p]);
    }
}


// GRU gate math with the by-products of the NEXT phase written in the same pass (one barrier less per layer and time step):
//   drop: hb2[row][j] = keep ? h_new * scale : 0   (the inter-layer dropout of nn.GRU: layer 1's input)
//   else: hb2[row][j] = h_new                        (layer 1's state copied into the [h1 | z] operand of the output projection)
__device__ __forceinline__ void gru_gates2(const float* gi, const float* gh, const float* bih, const float* bhh, float* hf, bf16* hb, int ldh,
                                           int r0, int R, float* save, bf16* hb2, int ldh2, const uint8_t* keep, float keep_scale, bool drop, int tid) {
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        const int row = idx / H, j = idx - row * H;
        const float* a = gi + row * GL;
        const float* b = gh + row * GL;
        const float r = sigm(a[j] + bih[j] + b[j] + bhh[j]);
        const float z = sigm(a[H + j] + bih[H + j] + b[H + j] + bhh[H + j]);
        const float ghn = b[2 * H + j] + bhh[2 * H + j];
        const float n = tanh_fast(a[2 * H + j] + bih[2 * H + j] + r * ghn);
        const float hp = hf[row * H + j];
        const float hn = (1.0f - z) * n + z * hp;
        hf[row * H + j] = hn;
        hb[row * ldh + j] = (bf16)hn;
        float v2 = hn;
        if (drop && keep && r0 + row < R) v2 = keep[(size_t)(r0 + row) * H + j] ? hn * keep_scale : 0.f;
        hb2[row * ldh2 + j] = (bf16)v2;
        if (save && r0 + row < R) {
            const size_t o = (size_t)(r0 + row) * H + j, pl = (size_t)R * H;
            save[o] = r; save[pl + o] = z; save[2 * pl + o] = n; save[3 * pl + o] = ghn; save[4 * pl + o] = hp;
        }
    }
}

// Forward of the text decoder with FIVE barriers per time step (the first version above has ten): the dropout of layer 0's output
// and the copy of layer 1's state ride in the gate passes, and the log-softmax / NLL / greedy argmax run on 16 lanes per row
// (shuffle reductions) in the same phase that embeds the fed-back token for the next step.
// KSX: k-steps of the [c_in | z] operand (kx / 32; 7 at n_latents = 100): the fragment registers of a GEMM are sized by it, and the
// two GEMMs of a phase have all their weight loads in flight together -- one step too many spilled to scratch in the hot loop
template <int KSX>
__global__ __launch_bounds__(NTHR) void text_decoder_fwd2_kernel(const TextDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DecLds L = dec_lds(smem, a.kx, a.kz);
    constexpr int LH = HP + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * TR, R = a.R, D = a.D;
    zero_bf(L.x0b, 2 * TR * L.LX + TR * L.LZ + 3 * TR * LH, tid);
    float* c_emb = L.cst; float* c_b = L.cst + 1200; float* c_h2o = L.cst + 2400; float* c_z2h = L.cst + 2416;
    for (int i = tid; i < 1200; i += NTHR) c_emb[i] = a.embed[i];
    for (int i = tid; i < 300; i += NTHR) {
        c_b[i] = a.l0.bih[i]; c_b[300 + i] = a.l0.bhh[i]; c_b[600 + i] = a.l1.bih[i]; c_b[900 + i] = a.l1.bhh[i];
    }
    if (tid < TXT_V) c_h2o[tid] = a.h2o_bias[tid];
    if (tid < H) c_z2h[tid] = a.z2h_bias[tid];
    __syncthreads();
    for (int idx = tid; idx < TR * D; idx += NTHR) {
        int row = idx / D, j = idx - row * D;
        float z = (r0 + row < R) ? a.z[(size_t)(r0 + row) * D + j] : 0.f;
        L.zf[row * 128 + j] = z;
        bf16 zb = (bf16)z;
        L.zb[row * L.LZ + j] = zb;
        L.x0b[row * L.LX + H + j] = zb;
        L.hzb[row * L.LX + H + j] = zb;
    }
    for (int idx = tid; idx < TR * H; idx += NTHR) {       // x0 of step 0: swish(embed(SOS))   (multimnist/utils.py:17, model.py:299)
        const int row = idx / H, j = idx - row * H;
        const float e = c_emb[10 * H + j];
        L.x0b[row * L.LX + j] = (bf16)(e / (1.0f + expf(-e)));
    }
    __syncthreads();
    save_tile(L.zb, L.LZ, a.kz, a.z_bf, a.kz, r0, R, tid);
    rowtile_gemm<4, 1>(L.zb, L.LZ, a.kz / 32, a.z2h, a.kz, 7, L.gi, GL, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        int row = idx / H, j = idx - row * H;
        float h = L.gi[row * GL + j] + c_z2h[j];
        L.h0f[idx] = h; L.h1f[idx] = h;
        L.h0b[row * LH + j] = (bf16)h; L.h1b[row * LH + j] = (bf16)h;
    }
    // ---- the GRU weights stay on the CU for the whole recurrence: W_ih0, W_hh0, W_ih1 as this wave's fragments in registers
    //      W_hh1: k-steps 0..2 as a copy of its fragment-major pack in LDS (19 tiles x 3 KB; all four would not fit the 160 KB), k-step 3 in registers
    static_assert(KSX == 7, "the resident form is sized for kx = 224 (n_latents = 100)");
    constexpr int NT = GL / 16;                               // 19 column tiles
    bf16* wl = reinterpret_cast<bf16*>(smem + dec_lds_bytes(KSX * 32, 128));
    for (int i = tid; i < NT * 3 * 64; i += NTHR) {           // LDS vector (tile, ks < 3, lane) <- pack vector (tile, ks, lane)
        const int nt = i / 192, rem = i - nt * 192;
        *reinterpret_cast<bf16x8*>(wl + (size_t)i * 8) = *reinterpret_cast<const bf16x8*>(a.l1.whh + ((size_t)nt * 256 + rem) * 8);
    }
    bf16x8 w_hh1_k3[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) w_hh1_k3[t] = *reinterpret_cast<const bf16x8*>(a.l1.whh + ((size_t)min(wave + t * NW, NT - 1) * 256 + 192 + lane) * 8);
    // (all four in registers spilled 53 VGPRs: W_hh0's fragments are re-read once per step instead -- issued behind layer 1's MFMAs,
    //  a whole phase ahead of their use, so the L2 latency is never waited for)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 w_ih0[3][KSX], w_ih1[3][4];
    load_wfrags<KSX, 3>(w_ih0, a.l0.wih, NT, wave, lane);
    load_wfrags<4, 3>(w_ih1, a.l1.wih, NT, wave, lane);
    __syncthreads();
    float nll[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t pl128 = (size_t)R * HP, plx = (size_t)R * a.kx;
    // softmax phase: thread (row, c) = (tid / 16, tid % 16) of the first 256 threads
    const int srow = (tid >> 4) & 15, sc = tid & 15;
    const bool srow_ok = r0 + srow < R;
    int nts = 0;
    auto stamp = [&]() { if (a.ts && blockIdx.x == 0 && tid == 0) a.ts[nts++] = __builtin_amdgcn_s_memrealtime(); };
    stamp();
#pragma unroll 1
    for (int i = 0; i < TXT_T; ++i) {
        save_tile(L.x0b, L.LX, a.kx, a.x0_bf ? a.x0_bf + i * plx : nullptr, a.kx, r0, R, tid);
        save_tile(L.h0b, LH, HP, a.h0p_bf ? a.h0p_bf + i * pl128 : nullptr, HP, r0, R, tid);
        save_tile(L.h1b, LH, HP, a.h1p_bf ? a.h1p_bf + i * pl128 : nullptr, HP, r0, R, tid);
        rowtile_gemm_w<KSX, 3>(L.x0b, L.LX, w_ih0, NT, L.gi, GL, wave, lane);
        rowtile_gemm<4, 3>(L.h0b, LH, HP / 32, a.l0.whh, HP, NT, L.gh, GL, wave, lane);
        __syncthreads();
        stamp();
        gru_gates2(L.gi, L.gh, c_b, c_b + 300, L.h0f, L.h0b, LH, r0, R, a.gates ? a.gates + (size_t)(i * 2 + 0) * 5 * R * H : nullptr,
                   L.midb, LH, a.keep ? a.keep + (size_t)i * R * H : nullptr, a.keep_scale, true, tid);
        __syncthreads();
        stamp();
        save_tile(L.midb, LH, HP, a.mid_bf ? a.mid_bf + i * pl128 : nullptr, HP, r0, R, tid);
        rowtile_gemm_w<4, 3>(L.midb, LH, w_ih1, NT, L.gi, GL, wave, lane);
        rowtile_gemm_lds3(L.h1b, LH, wl, w_hh1_k3, NT, L.gh, GL, wave, lane);
        __syncthreads();
        stamp();
        gru_gates2(L.gi, L.gh, c_b + 600, c_b + 900, L.h1f, L.h1b, LH, r0, R, a.gates ? a.gates + (size_t)(i * 2 + 1) * 5 * R * H : nullptr,
                   L.hzb, L.LX, nullptr, 1.f, false, tid);
        __syncthreads();
        stamp();
        save_tile(L.hzb, L.LX, a.kx, a.hz_bf ? a.hz_bf + i * plx : nullptr, a.kx, r0, R, tid);
        rowtile_gemm<KSX, 1>(L.hzb, L.LX, a.kx / 32, a.h2o, a.kx, 1, L.lg, 16, wave, lane);
        __syncthreads();
        stamp();
        if (tid < 256) {            // (whole waves: the shuffles below need every lane of a 16-lane group)
            const bool cv = sc < TXT_V;
            const float v = cv ? L.lg[srow * 16 + sc] + c_h2o[sc] : -INFINITY;
            float mx = v;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
            float se = cv ? expf(v - mx) : 0.f;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) se += __shfl_xor(se, o, 16);
            const float lp = v - (mx + logf(se));
            float bv = cv ? lp : -INFINITY; int best = sc;          // first maximum, like torch.max
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 16); const int oi = __shfl_xor(best, o, 16);
                if (ov > bv || (ov == bv && oi < best)) { bv = ov; best = oi; }
            }
            int tok = 10;
            if (srow_ok) {
                const size_t o = ((size_t)(r0 + srow) * TXT_T + i) * TXT_V;
                if (cv) a.words[o + sc] = lp;
                if (a.tokens_out && sc == 0) a.tokens_out[(size_t)(r0 + srow) * TXT_T + i] = best;
                if (a.target) {
                    const int pass = (r0 + srow) / a.rows_per_pass, b = (r0 + srow) - pass * a.rows_per_pass;
                    const int tg = (int)a.target[(size_t)b * TXT_T + i];
                    if (cv && sc == tg) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) nll[p] += (p == (pass & 3)) ? -lp : 0.f;
                    }
                    if (a.dwords && cv) a.dwords[o + sc] = (sc == tg) ? -a.nll_coef[pass & 3] : 0.f;
                }
                tok = a.force_tokens ? (int)a.force_tokens[(size_t)(r0 + srow) * TXT_T + i] : best;
            }
            // c_in of the next step = swish(embed(token))   (model.py:299-300); rows past the batch keep feeding SOS
            if (i + 1 < TXT_T)
                for (int j = sc; j < H; j += 16) {
                    const float e = c_emb[tok * H + j];
                    L.x0b[srow * L.LX + j] = (bf16)(e / (1.0f + expf(-e)));
                }
        }
        __syncthreads();
        stamp();
    }
    if (a.target && a.nll_sum) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float sum = wave_sum(nll[p]);
            if (lane == 0 && sum != 0.f) atomicAdd(a.nll_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + p, sum);
        }
    }
}

// ============================================================== text decoder backward
__global__ __launch_bounds__(NTHR) void text_decoder_bwd_kernel(const TextDecBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TextDecArgs& f = a.f;
    constexpr int LG = GK + 8, LO = 256;
    float* o1 = reinterpret_cast<float*>(smem);              // [16][256] gemm output
    float* dh0 = o1 + TR * LO;                               // [16][100] grad wrt layer-0 state
    float* dh1 = dh0 + TR * H;                               // [16][100]
    float* dhd = dh1 + TR * H;                               // [16][100]
    float* dzacc = dhd + TR * H;                             // [16][128]
    float* demb = dzacc + TR * 128;                          // [12][100]
    bf16* dgi = reinterpret_cast<bf16*>(demb + TXT_V * H);   // [16][328]
    bf16* dgh = dgi + TR * LG;                               // [16][328]
    bf16* dlg = dgh + TR * LG;                               // [16][40]
    bf16* dhb = dlg + TR * 40;                               // [16][136]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * TR, R = f.R, D = f.D, XI = H + D;
    const int nxt = round_up(XI, 16) / 16, ndt = round_up(D, 16) / 16;
    zero_bf(dgi, 2 * TR * LG + TR * 40 + TR * 136, tid);
    zero_f(dh0, 3 * TR * H + TR * 128 + TXT_V * H, tid);
    __syncthreads();
    float gb[4][2] = {};
    float gho = 0.f;
    for (int i = TXT_T - 1; i >= 0; --i) {
        // dlogit = dlp - softmax * sum(dlp)   (log_softmax backward)
        if (tid < TR) {
            const int row = tid;
            float d[TXT_V], s = 0.f;
            const bool ok = r0 + row < R && r0 + row < a.R_active;
            const size_t o = ((size_t)(r0 + row) * TXT_T + i) * TXT_V;
            for (int c = 0; c < TXT_V; ++c) { d[c] = ok ? a.dwords[o + c] : 0.f; s += d[c]; }
            for (int c = 0; c < TXT_V; ++c) {
                float v = ok ? d[c] - expf(f.words[o + c]) * s : 0.f;
                dlg[row * 40 + c] = (bf16)v;
                if (r0 + row < R) a.dlogit_bf[((size_t)i * R + r0 + row) * 16 + c] = (bf16)v;
            }
            if (r0 + row < R)
                for (int c = TXT_V; c < 16; ++c) a.dlogit_bf[((size_t)i * R + r0 + row) * 16 + c] = (bf16)0.f;
        }
        __syncthreads();
        if (tid < TXT_V) gho += colsum16(dlg, 40, tid);
        // d[h1 | z] = dlogit * W_h2o
        rowtile_gemm<1, 2>(dlg, 40, 1, f.h2oT, 32, nxt, o1, LO, wave, lane);
        __syncthreads();
        for (int idx = tid; idx < TR * XI; idx += NTHR) {
            int row = idx / XI, j = idx - row * XI;
            if (j < H) dh1[row * H + j] += o1[row * LO + j];
            else dzacc[row * 128 + j - H] += o1[row * LO + j];
        }
        __syncthreads();
        // ---- layer 1
        gru_gates_bwd(f.gates + (size_t)(i * 2 + 1) * 5 * R * H, r0, R, dh1, dgi, dgh, LG, dhd,
                      a.dgi1 + (size_t)i * R * GL, a.dgh1 + (size_t)i * R * GL, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + q * NTHR < TXT_G3) { gb[2][q] += colsum16(dgi, LG, tid + q * NTHR); gb[3][q] += colsum16(dgh, LG, tid + q * NTHR); }
        rowtile_gemm<10, 1>(dgh, LG, GK / 32, f.l1.whhT, GK, 7, o1, LO, wave, lane);
        __syncthreads();
        for (int idx = tid; idx < TR * H; idx += NTHR) dh1[idx] = dhd[idx] + o1[(idx / H) * LO + idx % H];
        __syncthreads();
        rowtile_gemm<10, 1>(dgi, LG, GK / 32, f.l1.wihT, GK, 7, o1, LO, wave, lane);     // d mid
        __syncthreads();
        for (int idx = tid; idx < TR * H; idx += NTHR) {
            int row = idx / H, j = idx - row * H;
            float v = o1[row * LO + j];
            if (f.keep && r0 + row < R) v = f.keep[((size_t)i * R + r0 + row) * H + j] ? v * f.keep_scale : 0.f;
            dh0[idx] += v;
        }
        __syncthreads();
        // ---- layer 0
        gru_gates_bwd(f.gates + (size_t)(i * 2 + 0) * 5 * R * H, r0, R, dh0, dgi, dgh, LG, dhd,
                      a.dgi0 + (size_t)i * R * GL, a.dgh0 + (size_t)i * R * GL, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + q * NTHR < TXT_G3) { gb[0][q] += colsum16(dgi, LG, tid + q * NTHR); gb[1][q] += colsum16(dgh, LG, tid + q * NTHR); }
        rowtile_gemm<10, 1>(dgh, LG, GK / 32, f.l0.whhT, GK, 7, o1, LO, wave, lane);
        __syncthreads();
        for (int idx = tid; idx < TR * H; idx += NTHR) dh0[idx] = dhd[idx] + o1[(idx / H) * LO + idx % H];
        __syncthreads();
        rowtile_gemm<10, 2>(dgi, LG, GK / 32, f.l0.wihT, GK, nxt, o1, LO, wave, lane);   // d[x_embed | z]
        __syncthreads();
        for (int idx = tid; idx < TR * XI; idx += NTHR) {
            int row = idx / XI, j = idx - row * XI;
            if (r0 + row >= R) continue;
            if (j < H) {
                // input token of step i: SOS for i == 0, else the fed-back token of step i-1
                int tok = 10;
                if (i > 0) tok = f.force_tokens ? (int)f.force_tokens[(size_t)(r0 + row) * TXT_T + i - 1]
                                                : (int)f.tokens_out[(size_t)(r0 + row) * TXT_T + i - 1];
                float e = f.embed[tok * H + j];
                float sg = 1.0f / (1.0f + expf(-e));
                atomicAdd(demb + tok * H + j, o1[row * LO + j] * sg * (1.0f + e * (1.0f - sg)));
            } else {
                dzacc[row * 128 + j - H] += o1[row * LO + j];
            }
        }
        __syncthreads();
    }
    // initial hidden state of both layers = z2h(z)
    for (int idx = tid; idx < TR * H; idx += NTHR) {
        int row = idx / H, j = idx - row * H;
        float v = dh0[idx] + dh1[idx];
        dhb[row * 136 + j] = (bf16)v;
        if (r0 + row < R) a.dhinit_bf[(size_t)(r0 + row) * 112 + j] = (bf16)v;
    }
    __syncthreads();
    if (tid < H) atomicAdd(a.g_z2h_bias + tid, colsum16(dhb, 136, tid));
    rowtile_gemm<4, 1>(dhb, 136, HP / 32, f.z2hT, HP, ndt, o1, LO, wave, lane);
    __syncthreads();
    for (int idx = tid; idx < TR * D; idx += NTHR) {
        int row = idx / D, j = idx - row * D;
        if (r0 + row < R) a.dz[(size_t)(r0 + row) * D + j] = dzacc[row * 128 + j] + o1[row * LO + j];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (tid + q * NTHR < TXT_G3) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) atomicAdd(a.g_b[k4] + tid + q * NTHR, gb[k4][q]);
        }
    if (tid < TXT_V) atomicAdd(a.g_h2o_bias + tid, gho);
    for (int idx = tid; idx < TXT_V * H; idx += NTHR)
        if (demb[idx] != 0.f) atomicAdd(a.g_embed + idx, demb[idx]);
}

template <typename K>
void set_lds_attr(K kernel, int bytes = 128 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

int launch_text_encoder_fwd(const TextEncArgs& a, hipStream_t s) {
    if (mmvae_knob("dbg_skip_text", 0)) return MMVAE_OK;          // measurement aid: the step without its text kernels
    MMVAE_REQUIRE(a.B >= 1 && 2 * a.D <= 256 && a.nh2p % 16 == 0, "text encoder: B=%d D=%d", a.B, a.D);
    size_t lds = (size_t)(2 * TR * GL + 2 * TR * H + 2400) * 4 + (size_t)3 * TR * (HP + 8) * 2;
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) set_lds_attr(text_encoder_fwd_kernel);
    hipLaunchKernelGGL(text_encoder_fwd_kernel, dim3(ceil_div(a.B, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("text_encoder_fwd");
}
int launch_text_encoder_bwd(const TextEncBwdArgs& a, hipStream_t s) {
    if (mmvae_knob("dbg_skip_text", 0)) return MMVAE_OK;          // measurement aid: the step without its text kernels
    const int K2 = round_up(2 * a.f.D, 32);
    size_t lds = (size_t)(TR * HP + 2 * TR * H + TXT_V * H) * 4 + (size_t)(2 * TR * (GK + 8) + TR * (K2 + 8)) * 2;
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) set_lds_attr(text_encoder_bwd_kernel);
    hipLaunchKernelGGL(text_encoder_bwd_kernel, dim3(ceil_div(a.f.B, TR)), dim3(NTHR), lds, s, a);
    return mmvae_check_launch("text_encoder_bwd");
}
int launch_text_decoder_fwd(const TextDecArgs& a_in, hipStream_t s) {
    const TextDecArgs& a = a_in;
    if (mmvae_knob("dbg_skip_text", 0)) return MMVAE_OK;          // measurement aid: the step without its text kernels
    MMVAE_REQUIRE(a.R >= 1 && a.D >= 1 && a.D <= 128 && a.kx == round_up(H + a.D, 32) && a.kz == round_up(a.D, 32) && a.kx <= 256,
                  "text decoder: R=%d D=%d kx=%d kz=%d", a.R, a.D, a.kx, a.kz);
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) { set_lds_attr(text_decoder_fwd_kernel); set_lds_attr(text_decoder_fwd2_kernel<7>, 160 * 1024); }
    // the weights-resident form is compiled for kx = 224, kz = 128 (n_latents = 97..100: the reference's 100); other sizes stream
    if (mmvae_knob("text_fwd2", 1) && a_in.kx == 224 && a_in.kz == 128 && a_in.l0.kih == 224 && a_in.l1.kih == HP) {
        TextDecArgs a = a_in;
        a.ts = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)mmvae_knob("txt_ts_hi", 0) << 32) | (unsigned)mmvae_knob("txt_ts_lo", 0));
        hipLaunchKernelGGL(text_decoder_fwd2_kernel<7>, dim3(ceil_div(a.R, TR)), dim3(NTHR), dec_lds_bytes(224, 128) + (size_t)(GL / 16) * 3 * 1024, s, a);
    } else hipLaunchKernelGGL(text_decoder_fwd_kernel, dim3(ceil_div(a.R, TR)), dim3(NTHR), dec_lds_bytes(a.kx, a.kz), s, a);
    return mmvae_check_launch("text_decoder_fwd");
}
int launch_text_decoder_bwd(const TextDecBwdArgs& a, hipStream_t s) {
    if (mmvae_knob("dbg_skip_text", 0)) return MMVAE_OK;          // measurement aid: the step without its text kernels
    MMVAE_REQUIRE(a.f.tokens_out != nullptr || a.f.force_tokens != nullptr, "text decoder bwd needs the token path");
    size_t lds = (size_t)(TR * 256 + 3 * TR * H + TR * 128 + TXT_V * H) * 4 + (size_t)(2 * TR * (GK + 8) + TR * 40 + TR * 136) * 2;
    static std::atomic<unsigned> once{0};
    if (mmvae_first_use_on_device(once)) set_lds_attr(text_decoder_bwd_kernel);
    MMVAE_LAUNCH(text_decoder_bwd_kernel, dim3(ceil_div(a.f.R, TR)), dim3(NTHR), lds, s, a);      // (the fused step joins on this kernel's completion event)
    return mmvae_check_launch("text_decoder_bwd");
}
