// Caption half of the COCO MMVAE (coco/model.py:219-312), fp32 on the fp32 MFMA GEMM of gemm_f32.hip.
//
// TextEncoder: bidirectional GRU(300 -> 200) over T = 102 GloVe vectors, last time step of both directions (the reverse
// direction has then seen only vector T-1 from h = 0), summed, Linear(200, 2D).
// TextDecoder: h = z2h(z) for both layers of a GRU(300+D -> 200, 2 layers, inter-layer dropout 0.1); the first input is
// GloVe('<s>'), every later input is the previous OUTPUT VECTOR (a differentiable feedback: backpropagation through time
// runs through the outputs as well as through the hidden states); output Linear(200+D -> 300).
//
// Why fp32: 102 dependent steps amplify operand rounding, and the reference's numbers are the parity target; the
// per-step GEMMs (3B x 600 x 400 at most) are latency-sized either way.  What is restructured, not approximated:
//   * the z-columns of the layer-0 input projection and of the output projection do not depend on the time step: they are
//     computed once per step of the optimizer (zi0 = z Wih0[:,300:]^T + b, zo = z Who[:,200:]^T + b) and added in the
//     GEMM epilogues; their gradients are one GEMM over the time-summed gate gradients;
//   * the encoder's input projection for all T steps is ONE GEMM (B*T rows);
//   * weight gradients whose operands share a [t][row] layout are ONE GEMM over K = T*rows (split-K over the grid).
// Saved per step for the backward pass: gates (r, z, n, W_hn h + b_hn) and hidden states.
#include "coco_plan.h"
#include "gemm_f32.h"

namespace {

constexpr int H = COCO_H, G = COCO_G, E = COCO_E, TPBT = 256;

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

struct GruFwd {
    const float* gi; long long ldgi;     // [rows][3H] input projection (+ bias), row stride ldgi
    const float* gh;                     // [rows][3H] hidden projection (+ bias), dense
    const float* hprev;                  // [rows][H]
    float* h;                            // [rows][H]
    float* sav;                          // [rows][4H] (r | z | n | gh_n) or null
    const uint8_t* keep; float scale;    // inter-layer dropout keep flags [rows][H] or null
    float* mid;                          // [rows][H] = dropout(h) or null
    int rows;
};
__global__ __launch_bounds__(TPBT) void gru_fwd_kernel(const GruFwd a) {
    const int i = blockIdx.x * TPBT + threadIdx.x;
    if (i >= a.rows * H) return;
    const int r = i / H, j = i - r * H;
    const float* gi = a.gi + (size_t)r * a.ldgi;
    const float* gh = a.gh + (size_t)r * G;
    const float rr = sigm(gi[j] + gh[j]);
    const float zz = sigm(gi[H + j] + gh[H + j]);
    const float ghn = gh[2 * H + j];
    const float nn = tanhf(gi[2 * H + j] + rr * ghn);
    const float hh = (1.0f - zz) * nn + zz * a.hprev[i];
    a.h[i] = hh;
    if (a.sav) {
        float* s = a.sav + (size_t)r * 4 * H;
        s[j] = rr; s[H + j] = zz; s[2 * H + j] = nn; s[3 * H + j] = ghn;
    }
    if (a.mid) a.mid[i] = a.keep ? hh * (float)a.keep[i] * a.scale : hh;
}
int gru_fwd(const GruFwd& a, hipStream_t s) {
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(ceil_div(a.rows * H, TPBT)), dim3(TPBT), 0, s, a);
    return mmvae_check_launch("gru_fwd");
}

struct GruBwd {
    const float* dh;                     // [rows][H] gradient wrt this step's h
    const float* dh2;                    // optional second addend (gradient wrt dropout(h) of the layer above)
    const uint8_t* keep; float scale;    // its dropout keep flags, or null
    const float* sav; const float* hprev;
    float* dgi; long long lddgi;         // [rows][3H] gradient wrt the input projection, row stride lddgi
    float* dgh;                          // [rows][3H] gradient wrt the hidden projection, dense
    float* dh_out;                       // [rows][H] = dh * z (the direct path to h_prev; may alias dh)
    int rows;
};
__global__ __launch_bounds__(TPBT) void gru_bwd_kernel(const GruBwd a) {
    const int i = blockIdx.x * TPBT + threadIdx.x;
    if (i >= a.rows * H) return;
    const int r = i / H, j = i - r * H;
    float d = a.dh[i];
    if (a.dh2) d += a.keep ? a.dh2[i] * (float)a.keep[i] * a.scale : a.dh2[i];
    const float* s = a.sav + (size_t)r * 4 * H;
    const float rr = s[j], zz = s[H + j], nn = s[2 * H + j], ghn = s[3 * H + j];
    const float dn = d * (1.0f - zz);
    const float dzz = d * (a.hprev[i] - nn);
    const float dpn = dn * (1.0f - nn * nn);
    const float dpz = dzz * zz * (1.0f - zz);
    const float dpr = dpn * ghn * rr * (1.0f - rr);
    float* gi = a.dgi + (size_t)r * a.lddgi;
    float* gh = a.dgh + (size_t)r * G;
    gi[j] = dpr; gi[H + j] = dpz; gi[2 * H + j] = dpn;
    gh[j] = dpr; gh[H + j] = dpz; gh[2 * H + j] = dpn * rr;
    a.dh_out[i] = d * zz;
}
int gru_bwd(const GruBwd& a, hipStream_t s) {
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(ceil_div(a.rows * H, TPBT)), dim3(TPBT), 0, s, a);
    return mmvae_check_launch("gru_bwd");
}

// ------------------------------------------------------------------ fused GRU layer step (forward)
// h' = GRU(x, h) in ONE launch: both projections on the fp32 MFMA and the gate math in the epilogue.  A workgroup owns 32
// rows x 16 hidden units; its accumulators are the four pre-activation pieces of those units (r and z: input + hidden
// parts summed in the same accumulator; n: input part and hidden part kept apart, n = tanh(i_n + r * h_n)).  The KS waves
// split the k-steps of both phases and are summed through LDS; operands go through buffer descriptors (hardware zero
// fill), U steps of loads are in flight before the first MFMA.  Replaces two GEMM launches, one gate kernel and the
// [rows][600] x 2 round trip of the projections through memory.
struct GruLayer {
    const float* x; long long ldx; int Kx;     // layer input [rows][Kx] (ldx = 0: one row for all), Kx = 0: none
    const float* Wih; long long ldwih;         // [3H][ldwih] input weights (column window applied by the caller)
    const float* gi_add; long long ldgi;       // [rows][3H] added to the input projection (precomputed part), or null
    const float* bih;                          // [3H] or null (already inside gi_add)
    const float* hprev;                        // [rows][H]
    const float* Whh; const float* bhh;        // [3H][H], [3H]
    float* h; float* sav;                      // [rows][H]; [rows][4H] (r | z | n | h_n) or null
    const uint8_t* keep; float scale; float* mid;   // inter-layer dropout of the output, or null
    int rows;
};
template <int KS, int U>
__global__ __launch_bounds__(KS * 64) void gru_layer_fwd_kernel(const GruLayer a) {
    __shared__ float red[KS > 1 ? (KS - 1) * 32 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 32, j0 = blockIdx.y * 16;
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool jok = j0 + fr < H;
    bool rok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) rok[t] = m0 + t * 16 + fr < a.rows;
    // one phase = one (A operand, weight matrix, K) triple; phase 0 feeds (r, z, i_n), phase 1 feeds (r, z, h_n)
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
        const float* A = phase == 0 ? a.x : a.hprev;
        const float* W = phase == 0 ? a.Wih : a.Whh;
        const int K = phase == 0 ? a.Kx : H;
        if (K == 0) continue;
        const unsigned lda = phase == 0 ? (unsigned)a.ldx * 4u : (unsigned)H * 4u;
        const unsigned ldw = phase == 0 ? (unsigned)a.ldwih * 4u : (unsigned)H * 4u;
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x7FFFFFFF, 0x00020000);
        unsigned arow[2], wrow[3];
#pragma unroll
        for (int t = 0; t < 2; ++t) arow[t] = (unsigned)(m0 + t * 16 + fr) * lda;
#pragma unroll
        for (int g = 0; g < 3; ++g) wrow[g] = (unsigned)(g * H + j0 + fr) * ldw;
        const int nst = (K + 15) / 16;
        const int st0 = nst * wave / KS, st1 = nst * (wave + 1) / KS;
        for (int st = st0; st < st1; st += U) {
            f32x4 av[U][2], bv[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kb = (st + u) * 16 + fq * 4;
                const bool kin = st + u < st1 && kb < K;          // K % 4 == 0: the 4 k are all inside or all outside
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    av[u][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (kin && rok[t]) ? arow[t] + (unsigned)kb * 4u : 0xFFFFFFFFu, 0, 0));
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    bv[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (kin && jok) ? wrow[g] + (unsigned)kb * 4u : 0xFFFFFFFFu, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t][q], bv[u][0][q], acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t][q], bv[u][1][q], acc[t][1], 0, 0, 0);
                        acc[t][2 + phase] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t][q], bv[u][2][q], acc[t][2 + phase], 0, 0, 0);
                    }
        }
    }
    if (KS > 1) {
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((wave - 1) * 32 + (t * 4 + c) * 4 + r) * 64 + lane] = acc[t][c][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < KS - 1; ++w)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][c][r] += red[(w * 32 + (t * 4 + c) * 4 + r) * 64 + lane];
    }
    const int j = j0 + fr;
    if (!jok) return;
    float br = a.bhh[j], bz = a.bhh[H + j], bin = 0.f;
    const float bhn = a.bhh[2 * H + j];
    if (a.bih) { br += a.bih[j]; bz += a.bih[H + j]; bin = a.bih[2 * H + j]; }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + t * 16 + fq * 4 + r;
            if (m >= a.rows) continue;
            float pr = acc[t][0][r] + br, pz = acc[t][1][r] + bz, pin = acc[t][2][r] + bin;
            if (a.gi_add) {
                const float* gi = a.gi_add + (size_t)m * a.ldgi;
                pr += gi[j]; pz += gi[H + j]; pin += gi[2 * H + j];
            }
            const float ghn = acc[t][3][r] + bhn;
            const float rr = sigm(pr), zz = sigm(pz);
            const float nn = tanhf(pin + rr * ghn);
            const size_t i = (size_t)m * H + j;
            const float hh = (1.0f - zz) * nn + zz * a.hprev[i];
            a.h[i] = hh;
            if (a.sav) {
                float* s = a.sav + (size_t)m * 4 * H;
                s[j] = rr; s[H + j] = zz; s[2 * H + j] = nn; s[3 * H + j] = ghn;
            }
            if (a.mid) a.mid[i] = a.keep ? hh * (float)a.keep[i] * a.scale : hh;
        }
}
int gru_layer_fwd(const GruLayer& a, hipStream_t s) {
    MMVAE_REQUIRE(a.Kx % 4 == 0 && (a.Kx == 0 || (a.ldx % 4 == 0 && a.ldwih % 4 == 0)), "gru_layer: Kx=%d ldx=%lld", a.Kx, a.ldx);
    MMVAE_REQUIRE((long long)a.rows * (a.ldx > H ? a.ldx : H) * 4 < 0x7FFFFFFFll, "gru_layer: operand spans more than 2 GiB");
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.Wih) | reinterpret_cast<uintptr_t>(a.hprev) |
                           reinterpret_cast<uintptr_t>(a.Whh)) & 15) == 0;
    MMVAE_REQUIRE(aligned, "gru_layer: operands must be 16-byte aligned");
    dim3 grid(ceil_div(a.rows, 32), ceil_div(H, 16));
    const int nst = ceil_div(a.Kx, 16) + ceil_div(H, 16);
    if ((int)(grid.x * grid.y) <= 512 || nst <= 16) hipLaunchKernelGGL((gru_layer_fwd_kernel<4, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gru_layer_fwd_kernel<2, 4>), grid, dim3(128), 0, s, a);
    return mmvae_check_launch("gru_layer_fwd");
}

__global__ __launch_bounds__(TPBT) void add2_kernel(const float* a, const float* b, long long n, float* out) {
    const long long i = (long long)blockIdx.x * TPBT + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
int add2(const float* a, const float* b, long long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n + TPBT - 1) / TPBT)), dim3(TPBT), 0, s, a, b, n, out);
    return mmvae_check_launch("add2");
}
// out[o][i] = sum_t in[(o*T + t)*inner + i]
__global__ __launch_bounds__(TPBT) void sum_mid_kernel(const float* in, long long outer, int T, long long inner, float* out) {
    const long long i = (long long)blockIdx.x * TPBT + threadIdx.x;
    if (i >= outer * inner) return;
    const long long o = i / inner, c = i - o * inner;
    const float* p = in + o * T * inner + c;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += p[(long long)t * inner];
    out[i] = acc;
}
int sum_mid(const float* in, long long outer, int T, long long inner, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_mid_kernel, dim3((unsigned)((outer * inner + TPBT - 1) / TPBT)), dim3(TPBT), 0, s, in, outer, T, inner, out);
    return mmvae_check_launch("sum_mid");
}

// grid.y = group; recon rows of group g compare with the same target
__global__ __launch_bounds__(TPBT) void mse3_kernel(const float* recon, const float* target, long long per_group, float c0, float c1,
                                                    float c2, float* loss_sum, float* dw) {
    const int g = blockIdx.y;
    const float coef = g == 0 ? c0 : (g == 1 ? c1 : c2);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * TPBT + threadIdx.x; i < per_group; i += (long long)gridDim.x * TPBT) {
        const float d = recon[g * per_group + i] - target[i];
        acc += d * d;
        if (dw) dw[g * per_group + i] = coef * 2.f * d;
    }
    acc = wave_sum(acc);
    if (loss_sum && (threadIdx.x & 63) == 0) atomicAdd(loss_sum + (blockIdx.x % MMVAE_LOSS_SLOTS) * 16 + 4 + g, acc);
}

// ---- GEMM shorthands over PyTorch-layout weights W[N][ldw] (a column window [c0, c0+K) of it) -------------------------
// y[M][N] (ldy) = x[M][K] (ldx) W[:, c0:c0+K]^T (+ bias) (+ addm)
F32Gemm mk_lin(const float* x, long long ldx, int M, const float* W, int N, int K, long long ldw, int c0, const float* bias,
               const float* addm, long long ldadd, float* y, long long ldy) {
    F32Gemm g{};
    g.A = x; g.a_rs = ldx; g.a_cs = 1; g.B = W + c0; g.b_rs = 1; g.b_cs = ldw;
    g.M = M; g.N = N; g.K = K; g.C = y; g.ldc = ldy; g.bias = bias; g.addm = addm; g.ldadd = ldadd;
    return g;
}
int lin(const float* x, long long ldx, int M, const float* W, int N, int K, long long ldw, int c0, const float* bias,
        const float* addm, long long ldadd, float* y, long long ldy, hipStream_t s) {
    return gemm_f32(mk_lin(x, ldx, M, W, N, K, ldw, c0, bias, addm, ldadd, y, ldy), s);
}
// dx[M][K] (lddx) (+)= dy[M][N] (lddy) W[:, c0:c0+K]
F32Gemm mk_dx(const float* dy, long long lddy, int M, const float* W, int N, int K, long long ldw, int c0, float* dx, long long lddx,
              int accumulate) {
    F32Gemm g{};
    g.A = dy; g.a_rs = lddy; g.a_cs = 1; g.B = W + c0; g.b_rs = ldw; g.b_cs = 1;
    g.M = M; g.N = K; g.K = N; g.C = dx; g.ldc = lddx; g.accumulate = accumulate;
    return g;
}
int lin_dx(const float* dy, long long lddy, int M, const float* W, int N, int K, long long ldw, int c0, float* dx, long long lddx,
           int accumulate, hipStream_t s) {
    return gemm_f32(mk_dx(dy, lddy, M, W, N, K, ldw, c0, dx, lddx, accumulate), s);
}
// dW[:, c0:c0+K] += dy[M][N]^T x[M][K]   (dW is zero at the start of the step; float atomics when K is split)
int lin_dw(const float* dy, long long lddy, const float* x, long long ldx, int M, float* dW, int N, int K, long long ldw, int c0,
           hipStream_t s) {
    F32Gemm g{};
    g.A = dy; g.a_rs = 1; g.a_cs = lddy; g.B = x; g.b_rs = ldx; g.b_cs = 1;
    g.M = N; g.N = K; g.K = M; g.C = dW + c0; g.ldc = ldw; g.accumulate = 1;
    const int tiles = ceil_div(N, 32) * ceil_div(K, 32);
    int ks = 1;
    while (tiles * ks < 1024 && M / (ks * 2) >= 512 && ks < 64) ks *= 2;
    g.ksplit = ks;
    return gemm_f32(g, s);
}

}  // namespace

void coco_text_build(CocoPlan& P) {
    auto gru = [&](CocoGru& g, const std::string& n, const char* sfx) {
        g.wih = off(P, n + ".weight_ih_" + sfx); g.whh = off(P, n + ".weight_hh_" + sfx);
        g.bih = off(P, n + ".bias_ih_" + sfx); g.bhh = off(P, n + ".bias_hh_" + sfx);
    };
    gru(P.te_f, "text_encoder.gru", "l0"); gru(P.te_r, "text_encoder.gru", "l0_reverse");
    gru(P.td0, "text_decoder.gru", "l0"); gru(P.td1, "text_decoder.gru", "l1");
    P.te_h2p_w = off(P, "text_encoder.h2p.weight"); P.te_h2p_b = off(P, "text_encoder.h2p.bias");
    P.td_z2h_w = off(P, "text_decoder.z2h.weight"); P.td_z2h_b = off(P, "text_decoder.z2h.bias");
    P.td_h2o_w = off(P, "text_decoder.h2o.weight"); P.td_h2o_b = off(P, "text_decoder.h2o.bias");
    // ---- bf16 persistent decoder (coco_text_bf16.hip): weights packed MFMA-fragment-major (PackDesc::frag), forward and
    // transposed forms
    const int D = P.D, in0 = E + D, ino = H + D;
    static const bool fp32_text = getenv("MMVAE_COCO_TEXT_FP32") != nullptr;
    P.text_bf16 = !fp32_text;
    auto fragd = [&](PackDesc d) { d.frag = 1; return P.pk.add(d); };
    auto fwdp = [&](long long w, int N, int K, int ld, int Npad, int Kpad) { return fragd(pack_dense(w, N, K, Npad, Kpad, ld, 1)); };
    auto trp = [&](long long w, int Nout, int Kin, int ld, int Npad, int Kpad) { return fragd(pack_dense(w, Nout, Kin, Npad, Kpad, 1, ld)); };
    // the caption ENCODER's packs first: the step's critical chain starts with them (pack ranges: CocoPlan::pk_textdec_begin)
    P.tb_e_hh = fwdp(P.te_f.whh, G, H, H, CTB_GP, CTB_HP);      // caption encoder, forward direction
    P.tb_e_hhT = trp(P.te_f.whh, H, G, H, 208, CTB_GP);
    // weight_ih of the encoder as the ROW operand of the transposed input projection gi^T[600][T*B] = W_ih x^T (row-major)
    P.tb_e_ihA = P.pk.add(pack_dense(P.te_f.wih, G, E, round_up(G, 128), CTB_XP, E, 1));
    for (int g3 = 0; g3 < 3; ++g3)                               // per-gate copies for the weight-resident forward kernel
        P.tb_e_hhg[g3] = fwdp(P.te_f.whh + (long long)g3 * H * H, H, H, H, 208, CTB_HP);
    P.pk_textdec_begin = (int)P.pk.d.size();
    P.tb_ih0 = fwdp(P.td0.wih, G, E, in0, CTB_GP, CTB_XP);      // columns 0..299 of weight_ih_l0 (the word vector part)
    P.tb_hh0 = fwdp(P.td0.whh, G, H, H, CTB_GP, CTB_HP);
    P.tb_ih1 = fwdp(P.td1.wih, G, H, H, CTB_GP, CTB_HP);
    P.tb_hh1 = fwdp(P.td1.whh, G, H, H, CTB_GP, CTB_HP);
    P.tb_ho = fwdp(P.td_h2o_w, E, H, ino, CTB_EP, CTB_HP);      // columns 0..199 of h2o (the hidden part)
    P.tb_hoT = trp(P.td_h2o_w, H, E, ino, 208, CTB_XP);         // [j][e] = h2o[e][j]
    P.tb_ih1T = trp(P.td1.wih, H, G, H, 208, CTB_GP);
    P.tb_hh1T = trp(P.td1.whh, H, G, H, 208, CTB_GP);
    P.tb_hh0T = trp(P.td0.whh, H, G, H, 208, CTB_GP);
    P.tb_ih0T = trp(P.td0.wih, E, G, in0, CTB_EP, CTB_GP);      // [e][g] = weight_ih_l0[g][e]
    // per-gate copies of the decoder GRUs (cluster form): the three gates of a matrix back to back
    for (int g3 = 0; g3 < 3; ++g3) P.tb_g_ih0[g3] = fwdp(P.td0.wih + (long long)g3 * H * in0, H, E, in0, 208, CTB_XP);
    for (int g3 = 0; g3 < 3; ++g3) P.tb_g_hh0[g3] = fwdp(P.td0.whh + (long long)g3 * H * H, H, H, H, 208, CTB_HP);
    for (int g3 = 0; g3 < 3; ++g3) P.tb_g_ih1[g3] = fwdp(P.td1.wih + (long long)g3 * H * H, H, H, H, 208, CTB_HP);
    for (int g3 = 0; g3 < 3; ++g3) P.tb_g_hh1[g3] = fwdp(P.td1.whh + (long long)g3 * H * H, H, H, H, 208, CTB_HP);
    // W_ih0[:, :300]^T row-major ([300][600]): the batched GEMM dgi0[t+1] W_ih0x behind the composed BPTT kernel
    P.tb_ih0xT_rm = P.pk.add(pack_dense(P.td0.wih, E, G, npad_for(E), round_up(G, 64), 1, in0));
    // packed gradients [round64(N)][Kpad of the wgrad operand] -> scattered back by the unpack kernel
    // bias >= 0: the saved operand carries 1.0 in column K, so column K of the packed gradient is the bias gradient
    auto gkp = [&](long long w, int N, int K, int ld, int Kc, long long bias) {
        PackDesc d = pack_dense(w, N, K, round_up(N, 64), round_up(Kc, 64), ld, 1);
        if (bias >= 0) { d.bias_off = bias; d.b_nhi = 0; d.b_nlo = 1; }
        return P.gk.add(d);
    };
    P.tg_ih0 = gkp(P.td0.wih, G, E, in0, CTB_XP, -1);           // (b_ih of layer 0 sees the time sum: with the z-terms)
    P.tg_hh0 = gkp(P.td0.whh, G, H, H, CTB_HP, P.td0.bhh);
    P.tg_ih1 = gkp(P.td1.wih, G, H, H, CTB_HP, P.td1.bih);
    P.tg_hh1 = gkp(P.td1.whh, G, H, H, CTB_HP, P.td1.bhh);
    P.tg_ho = gkp(P.td_h2o_w, E, H, ino, CTB_HP, -1);
    P.tg_e_ih = gkp(P.te_f.wih, G, E, E, CTB_XP, P.te_f.bih);
    P.tg_e_hh = gkp(P.te_f.whh, G, H, H, CTB_HP, P.te_f.bhh);
}

void coco_text_carve(CocoPlan& P, Workspace& ws) {
    CocoPlan::W& w = P.w;
    const size_t B = P.B, R = (size_t)P.carve_passes * B, T = P.T, D = P.D;
    w.te_gi = ws.take<float>(B * T * G); w.te_gh = ws.take<float>(B * G); w.te_h = ws.take<float>(T * B * H);
    w.te_sav = ws.take<float>(T * B * 4 * H); w.te_gi_r = ws.take<float>(B * G); w.te_sav_r = ws.take<float>(B * 4 * H);
    w.te_hb = ws.take<float>(B * H); w.te_sum = ws.take<float>(B * H); w.txtout = ws.take<float>(B * 2 * D);
    w.d_txtout = ws.take<float>(B * 2 * D); w.te_dsum = ws.take<float>(B * H); w.te_dh = ws.take<float>(B * H);
    w.te_dgi = ws.take<float>(B * T * G); w.te_dgh = ws.take<float>(T * B * G); w.te_dgi_r = ws.take<float>(B * G);
    w.td_zi0 = ws.take<float>(R * G); w.td_zo = ws.take<float>(R * E); w.td_gi = ws.take<float>(R * G); w.td_gh = ws.take<float>(R * G);
    w.td_h0 = ws.take<float>((T + 1) * R * H); w.td_h1 = ws.take<float>((T + 1) * R * H); w.td_mid = ws.take<float>(T * R * H);
    w.td_sav0 = ws.take<float>(T * R * 4 * H); w.td_sav1 = ws.take<float>(T * R * 4 * H); w.td_recon = ws.take<float>(R * T * E);
    w.td_dw = ws.take<float>(R * T * E);
    w.td_dgi0 = ws.take<float>(T * R * G); w.td_dgh0 = ws.take<float>(T * R * G);
    w.td_dgi1 = ws.take<float>(T * R * G); w.td_dgh1 = ws.take<float>(T * R * G);
    w.td_dmid = ws.take<float>(R * H); w.td_dzi0 = ws.take<float>(R * G); w.td_dwsum = ws.take<float>(R * E); w.td_dhinit = ws.take<float>(R * H);
    if (P.text_bf16) {
        w.tb_x = ws.take<bf16>(T * R * CTB_XP); w.tb_h0 = ws.take<bf16>((T + 1) * R * CTB_HP); w.tb_mid = ws.take<bf16>(T * R * CTB_HP);
        w.tb_h1 = ws.take<bf16>((T + 1) * R * CTB_HP); w.tb_dout = ws.take<bf16>(T * R * CTB_EP);
        w.tb_dgi0 = ws.take<bf16>(T * R * CTB_GP); w.tb_dgh0 = ws.take<bf16>(T * R * CTB_GP);
        w.tb_dgi1 = ws.take<bf16>(T * R * CTB_GP); w.tb_dgh1 = ws.take<bf16>(T * R * CTB_GP);
        w.te_giT = ws.take<float>(B * T * G); w.te_hlast = ws.take<float>(B * H);
        // cluster exchange buffers (coco_text_bf16.hip CLF_BYTES / CLB_BYTES per row block) + a timeout word each
        w.cl_bytes = ((R + 15) / 16) * (size_t)(16 * 200 * 4 + 16 * 200 * 2 + 16 * 300 * 2 + 256) + 64;
        w.cl_xchg = ws.take<char>(w.cl_bytes);
        w.clb_bytes = ((R + 15) / 16) * (size_t)(2 * 16 * 200 * 8 + 16 * 300 * 2 + 256) + 64;
        w.clb_xchg = ws.take<char>(w.clb_bytes);
        w.te_xb = ws.take<bf16>((size_t)round_up((int)(T * B), 128) * CTB_XP); w.te_hb_all = ws.take<bf16>(T * B * CTB_HP);
        w.te_dgi_b = ws.take<bf16>(T * B * CTB_GP); w.te_dgh_b = ws.take<bf16>(T * B * CTB_GP);
        w.tb_comb = ws.take<bf16>(3 * 208 * CTB_HP); w.tb_combT = ws.take<bf16>(208 * CTB_GP);
        w.td_sosv = ws.take<float>(G); w.td_zi0p = ws.take<float>(R * G);
        w.tb_dw16 = ws.take<bf16>(R * T * CTB_XP); w.td_dzi1 = ws.take<float>(R * G);
        w.td_wz = ws.take<float>((size_t)G * D); w.td_bz = ws.take<float>(G);
    }
}

// ranks per 16-row block of the caption decoder's persistent launches (cluster form), 0 / 1: one workgroup per block
static int coco_dec_cluster(int R) {
    int Pc = mmvae_knob("coco_cluster", 8);                     // (A/B aid, mmvae_debug_set: tools/coco_cluster_check.py)
    const int nblk_pad = ((R + 15) / 16 + 7) / 8 * 8;
    // every rank of every row block must be resident at once (they spin on each other), next to the image half and the weight
    // gradients on the other streams: 7/8 of the device's CUs at most, else fewer ranks per block, else the one-workgroup kernels
    const int limit = mmvae_cu_count() * 7 / 8;
    while (Pc > 1 && nblk_pad * Pc > limit) Pc /= 2;
    return (Pc == 4 || Pc == 8) ? Pc : 0;
}

// composed form of the cluster-of-8 decoder (two exchanges per step): knob coco_no_comb keeps the three-exchange kernels (A/B aid)
static bool coco_dec_composed(const CocoPlan& P, int Pc) {
    const bool off = mmvae_knob("coco_no_comb", 0) != 0;
    return Pc == 8 && !off && H + P.D <= 320;                   // (coco_comb_kernel: one thread per column of [W_ho | its z part])
}
bool coco_text_dec_composed(const CocoPlan& P, int R) { return P.text_bf16 && coco_dec_composed(P, coco_dec_cluster(R)); }
int coco_text_dec_prepare(CocoPlan& P, const float* sos, hipStream_t s) {
    if (!P.text_bf16 || P.comb_fresh || H + P.D > 320) return MMVAE_OK;
    const float* p = P.buf.params;
    MMVAE_TRY(launch_coco_comb(p + P.td0.wih, p + P.td0.bih, p + P.td_h2o_w, p + P.td_h2o_b, P.D, sos, P.w.tb_comb, P.w.tb_combT, P.w.td_sosv,
                               P.w.td_wz, P.w.td_bz, s));
    P.comb_fresh = true;
    return MMVAE_OK;
}

// the weight-resident encoder kernels (coco_text_bf16.hip) need 4-row vectors of the batch
static bool coco_enc_resident(const CocoPlan& P) {
    const bool streamed = mmvae_knob("coco_enc_streamed", 0) != 0;          // A/B aid: the weight-streaming kernels
    return P.text_bf16 && !streamed && P.B % 4 == 0;
}

// ================================================================== caption encoder (coco/model.py:236-245)
int coco_text_enc_fwd(CocoPlan& P, const float* text, int save, float* out, hipStream_t s, bool bf16_path, const hipStream_t* side) {
    CocoPlan::W& w = P.w;
    const int B = P.B, T = P.T, D2 = 2 * P.D;
    const float* p = P.buf.params;
    const bool res_path = bf16_path && P.text_bf16 && coco_enc_resident(P);
    // reverse direction: its output at the last position is its FIRST step (input T-1, h = 0) -- three small launches that need
    // nothing of the forward recurrence.  With a side stream (ordered behind the step's prologue) they leave the chain.
    const hipStream_t sr = side ? *side : s;        // (a pointer: the default stream's handle is null)
    auto reverse_dir = [&]() -> int {
        MMVAE_TRY(lin(text + (size_t)(T - 1) * E, (long long)T * E, B, p + P.te_r.wih, G, E, E, 0, p + P.te_r.bih, nullptr, 0, w.te_gi_r, G, sr));
        MMVAE_TRY(lin(w.zeros_h, H, B, p + P.te_r.whh, G, H, H, 0, p + P.te_r.bhh, nullptr, 0, w.te_gh, G, sr));
        GruFwd a{};
        a.gi = w.te_gi_r; a.ldgi = G; a.gh = w.te_gh; a.hprev = w.zeros_h; a.h = w.te_hb; a.sav = save ? w.te_sav_r : nullptr; a.rows = B;
        return gru_fwd(a, sr);
    };
    if (sr != s) MMVAE_TRY(reverse_dir());
    if (res_path) {
        // input projection of every time step at once, TRANSPOSED: gi^T[600][T*B] = W_ih (rows) x captions^T, one bf16 GEMM with
        // the captions in the [t][row] bf16 layout the weight gradient needs anyway as its "weight" operand (b_ih is added in
        // fp32 by the recurrence kernel).  Replaces an fp32 GEMM (83 us at B=128, 590 us at B=1024) + a transpose.
        MMVAE_TRY(launch_coco_text_tb(text, B, T, CTB_XP, w.te_xb, s));
        GatherPlan pl = dense_plan(G, CTB_XP, CTB_XP, T * B);
        GemmParams g = gemm_of(P, pl, &P.tb_e_ihA, 1, G);
        g.c.A = P.buf.packed + P.pk.d[P.tb_e_ihA].dst_off;
        g.cls[0].Wp = w.te_xb; g.cls[0].Kpad = CTB_XP; g.npad = round_up(T * B, 128);
        g.out_f = w.te_giT; g.ldo = T * B;
        MMVAE_TRY(launch_gemm_gather(g, s));
    } else {
        // input projection of every time step at once: rows (b, t)
        MMVAE_TRY(lin(text, E, B * T, p + P.te_f.wih, G, E, E, 0, p + P.te_f.bih, nullptr, 0, w.te_gi, G, s));
    }
    if (bf16_path && P.text_bf16) {      // the recurrence in ONE persistent launch (coco_text_bf16.hip)
        CocoEncFwdArgs a{};
        const bool res = coco_enc_resident(P);
        a.B = B; a.T = T; a.gi = w.te_gi; a.bhh = p + P.te_f.bhh;
        a.resident = res;
        a.w_hh = P.buf.packed + P.pk.d[res ? P.tb_e_hhg[0] : P.tb_e_hh].dst_off;
        if (res) {      // the resident kernels read / write with the batch row fastest: [T][600][B] and friends
            MMVAE_REQUIRE(P.pk.d[P.tb_e_hhg[1]].dst_off == P.pk.d[P.tb_e_hhg[0]].dst_off + 208ll * CTB_HP &&
                          P.pk.d[P.tb_e_hhg[2]].dst_off == P.pk.d[P.tb_e_hhg[0]].dst_off + 2 * 208ll * CTB_HP, "per-gate packs not contiguous");
            a.gi = w.te_giT; a.bih = p + P.te_f.bih; a.h_last = w.te_hlast;
        }
        a.h_all = w.te_h;
        if (save) { a.sav = w.te_sav; a.hb_all = w.te_hb_all; }
        MMVAE_TRY(launch_coco_enc_fwd(a, s));
    } else
    for (int t = 0; t < T; ++t) {
        const float* hp = t == 0 ? w.zeros_h : w.te_h + (size_t)(t - 1) * B * H;
        GruLayer a{};
        a.gi_add = w.te_gi + (size_t)t * G; a.ldgi = (long long)T * G; a.hprev = hp; a.Whh = p + P.te_f.whh; a.bhh = p + P.te_f.bhh;
        a.h = w.te_h + (size_t)t * B * H; a.sav = save ? w.te_sav + (size_t)t * B * 4 * H : nullptr; a.rows = B;
        MMVAE_TRY(gru_layer_fwd(a, s));
    }
    if (sr == s) MMVAE_TRY(reverse_dir());
    else MMVAE_TRY(edge(P, sr, s));
    const float* h_last = bf16_path && P.text_bf16 && coco_enc_resident(P) ? w.te_hlast : w.te_h + (size_t)(T - 1) * B * H;
    MMVAE_TRY(add2(h_last, w.te_hb, (long long)B * H, w.te_sum, s));
    return lin(w.te_sum, H, B, p + P.te_h2p_w, D2, H, H, 0, p + P.te_h2p_b, nullptr, 0, out, D2, s);
}

int coco_text_enc_bwd(CocoPlan& P, const float* text, const float* d_out, hipStream_t s, hipStream_t sw, bool bf16_path) {
    CocoPlan::W& w = P.w;
    const int B = P.B, T = P.T, D2 = 2 * P.D;
    const float* p = P.buf.params;
    float* g = P.buf.grads;
    const bool bf = bf16_path && P.text_bf16;
    const bool a_res_done = bf && coco_enc_resident(P);
    hipStream_t so = bf ? sw : s;        // where everything that only feeds the optimizer goes
    MMVAE_TRY(lin_dx(d_out, D2, B, p + P.te_h2p_w, D2, H, H, 0, w.te_dsum, H, 0, s));
    if (bf) {       // BPTT of the forward direction in one launch; it is the critical chain, everything else goes behind it
        if (so != s) MMVAE_TRY(edge(P, s, so));
        CocoEncBwdArgs a{};
        a.B = B; a.T = T; a.dh_init = w.te_dsum; a.sav = w.te_sav; a.h_all = w.te_h;
        a.w_hhT = P.buf.packed + P.pk.d[P.tb_e_hhT].dst_off; a.dgi_b = w.te_dgi_b; a.dgh_b = w.te_dgh_b;
        a.resident = coco_enc_resident(P);
        MMVAE_TRY(launch_coco_enc_bwd(a, s));
        MMVAE_TRY(coco_text_dec_wgrads(P, so));      // the caption decoder's weight gradients, deferred to here by the step
    }
    MMVAE_TRY(launch_colsum_f32(d_out, B, D2, g + P.te_h2p_b, so));
    MMVAE_TRY(lin_dw(d_out, D2, w.te_sum, H, B, g + P.te_h2p_w, D2, H, H, 0, so));
    {   // reverse direction (one step from h = 0: no hidden-weight gradient, only its bias)
        GruBwd a{};
        a.dh = w.te_dsum; a.sav = w.te_sav_r; a.hprev = w.zeros_h; a.dgi = w.te_dgi_r; a.lddgi = G; a.dgh = w.te_gh; a.dh_out = w.te_dh; a.rows = B;
        MMVAE_TRY(gru_bwd(a, so));
        MMVAE_TRY(launch_colsum_f32(w.te_dgi_r, B, G, g + P.te_r.bih, so));
        MMVAE_TRY(launch_colsum_f32(w.te_gh, B, G, g + P.te_r.bhh, so));
        MMVAE_TRY(lin_dw(w.te_dgi_r, G, text + (size_t)(T - 1) * E, (long long)T * E, B, g + P.te_r.wih, G, E, E, 0, so));
    }
    if (bf) {       // weight gradients of the forward direction as batched bf16 GEMMs over all T*B rows
        if (!a_res_done) MMVAE_TRY(launch_coco_text_tb(text, B, T, CTB_XP, w.te_xb, so));    // (the resident forward path made it already)
        if (so != s) MMVAE_TRY(edge(P, s, so));
        WgradParams list[2];
        int n = 0;
        auto wg = [&](int gidx, const bf16* Pm, const bf16* Gm, int C, int rows) {
            GatherPlan pl = dense_plan(rows, C, C, G);
            WgradParams q = wgrad_of(P, pl, &gidx, 1, rows);
            q.c.A = Gm; q.P = Pm; q.ldp = CTB_GP;
            list[n++] = q;
        };
        // pairs (dgi[t], x[t]) and (dgh[t], h[t-1]); column 300 / 200 of the second operand is 1.0: b_ih, b_hh ride along
        wg(P.tg_e_ih, w.te_dgi_b, w.te_xb, CTB_XP, T * B);
        wg(P.tg_e_hh, w.te_dgh_b, w.te_hb_all, CTB_HP, T * B);
        MMVAE_TRY(launch_wgrad_group(list, n, so, &P.slab));
        return launch_wgrad_reduce(&P.slab, so, true);
    }
    // forward direction: backpropagation through time
    hipMemcpyAsync(w.te_dh, w.te_dsum, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, s);
    for (int t = T - 1; t >= 0; --t) {
        const float* hp = t == 0 ? w.zeros_h : w.te_h + (size_t)(t - 1) * B * H;
        GruBwd a{};
        a.dh = w.te_dh; a.sav = w.te_sav + (size_t)t * B * 4 * H; a.hprev = hp;
        a.dgi = w.te_dgi + (size_t)t * G; a.lddgi = (long long)T * G; a.dgh = w.te_dgh + (size_t)t * B * G; a.dh_out = w.te_dh; a.rows = B;
        MMVAE_TRY(gru_bwd(a, s));
        if (t > 0) MMVAE_TRY(lin_dx(w.te_dgh + (size_t)t * B * G, G, B, p + P.te_f.whh, G, H, H, 0, w.te_dh, H, 1, s));
    }
    MMVAE_TRY(lin_dw(w.te_dgi, G, text, E, B * T, g + P.te_f.wih, G, E, E, 0, s));
    MMVAE_TRY(launch_colsum_f32(w.te_dgi, B * T, G, g + P.te_f.bih, s));
    MMVAE_TRY(launch_colsum_f32(w.te_dgh, B * T, G, g + P.te_f.bhh, s));
    if (T > 1)      // pairs (dgh[t], h[t-1]) for t = 1..T-1, both laid out [t][b]
        MMVAE_TRY(lin_dw(w.te_dgh + (size_t)B * G, G, w.te_h, H, B * (T - 1), g + P.te_f.whh, G, H, H, 0, s));
    return MMVAE_OK;
}

// ================================================================== caption decoder (coco/model.py:266-312)
int coco_text_dec_fwd(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, int save, float* sentence,
                      hipStream_t s, bool bf16_path, const CocoMseFuse* mse) {
    P.mse_fused = false;
    CocoPlan::W& w = P.w;
    const int R = groups * P.B, T = P.T, D = P.D, in0 = E + D, ino = H + D;
    const float* p = P.buf.params;
    const size_t RH = (size_t)R * H;
    MMVAE_REQUIRE(sos != nullptr, "coco text decoder: the '<s>' vector must be given");
    // time-invariant z terms
    MMVAE_TRY(lin(z, D, R, p + P.td0.wih, G, D, in0, E, p + P.td0.bih, nullptr, 0, w.td_zi0, G, s));
    MMVAE_TRY(lin(z, D, R, p + P.td_h2o_w, E, D, ino, H, p + P.td_h2o_b, nullptr, 0, w.td_zo, E, s));
    MMVAE_TRY(lin(z, D, R, p + P.td_z2h_w, H, D, D, 0, p + P.td_z2h_b, nullptr, 0, w.td_h0, H, s));
    const bool composed = bf16_path && P.text_bf16 && coco_dec_composed(P, coco_dec_cluster(R));
    if (!composed) hipMemcpyAsync(w.td_h1, w.td_h0, RH * sizeof(float), hipMemcpyDeviceToDevice, s);   // (the composed kernel writes slice 0 of h1 itself)
    const float scale = 1.f / (1.f - DROP_P);
    if (bf16_path && P.text_bf16) {      // the whole recurrence in ONE persistent launch (coco_text_bf16.hip)
        CocoDecFwdArgs a{};
        a.R = R; a.T = T; a.hinit = w.td_h0; a.zi0 = w.td_zi0; a.zo = w.td_zo; a.sos = sos; a.keep = keep; a.keep_scale = scale;
        auto pw = [&](int i) { return P.buf.packed + P.pk.d[i].dst_off; };
        a.w_ih0 = pw(P.tb_ih0); a.w_hh0 = pw(P.tb_hh0); a.w_ih1 = pw(P.tb_ih1); a.w_hh1 = pw(P.tb_hh1); a.w_ho = pw(P.tb_ho);
        a.bhh0 = p + P.td0.bhh; a.bih1 = p + P.td1.bih; a.bhh1 = p + P.td1.bhh;
        a.sentence = sentence;
        {   // cluster form: P workgroups per 16-row block when the row blocks leave most of the chip idle
            const int Pc = coco_dec_cluster(R);
            if (Pc == 4 || Pc == 8) {
                a.cluster = Pc;
                a.wg_ih0 = pw(P.tb_g_ih0[0]); a.wg_hh0 = pw(P.tb_g_hh0[0]); a.wg_ih1 = pw(P.tb_g_ih1[0]); a.wg_hh1 = pw(P.tb_g_hh1[0]);
                for (int g3 = 1; g3 < 3; ++g3)
                    MMVAE_REQUIRE(P.pk.d[P.tb_g_ih0[g3]].dst_off == P.pk.d[P.tb_g_ih0[0]].dst_off + (long long)g3 * 208 * CTB_XP &&
                                  P.pk.d[P.tb_g_hh0[g3]].dst_off == P.pk.d[P.tb_g_hh0[0]].dst_off + (long long)g3 * 208 * CTB_HP &&
                                  P.pk.d[P.tb_g_ih1[g3]].dst_off == P.pk.d[P.tb_g_ih1[0]].dst_off + (long long)g3 * 208 * CTB_HP &&
                                  P.pk.d[P.tb_g_hh1[g3]].dst_off == P.pk.d[P.tb_g_hh1[0]].dst_off + (long long)g3 * 208 * CTB_HP,
                                  "per-gate decoder packs not contiguous");
                a.cl_xchg = reinterpret_cast<unsigned long long*>(w.cl_xchg);
                a.cl_timeout = reinterpret_cast<unsigned*>(w.cl_xchg + w.cl_bytes - 64);
                P.cl_alarm_f = a.cl_timeout;
                MMVAE_TRY(launch_fill_zero(w.cl_xchg, w.cl_bytes, s));      // flags and the timeout word: zero before EVERY launch
                if (coco_dec_composed(P, Pc)) {
                    MMVAE_TRY(coco_text_dec_prepare(P, sos, s));            // (no-op when the step made W_comb already)
                    // zi0p = zi0 + zo W_ih0x^T, what the composed input projection adds to W_comb h1, straight from z through the
                    // composed z-weights the preparation left: W_ih0z + W_ih0x W_hoz and b_ih + W_ih0x b_ho
                    MMVAE_TRY(lin(z, D, R, w.td_wz, G, D, D, 0, w.td_bz, nullptr, 0, w.td_zi0p, G, s));
                    a.wg_comb = w.tb_comb; a.zi0p = w.td_zi0p; a.sosv = w.td_sosv;
                    if (mse && !mmvae_knob("coco_no_mse_fuse", 0)) {        // (A/B aid)
                        a.mse_target = mse->target; a.mse_B = P.B; a.mse_loss = mse->loss_sum; a.mse_dw = mse->dw; a.mse_dw16 = mse->dw16;
                        for (int g3 = 0; g3 < 3; ++g3) a.mse_coef[g3] = g3 < groups ? mse->coef[g3] : 0.f;
                        P.mse_fused = true;
                    }
                }
            }
        }
        if (save) {
            a.h0_all = w.td_h0; a.h1_all = w.td_h1; a.sav0 = w.td_sav0; a.sav1 = w.td_sav1;
            a.xb_all = w.tb_x; a.h0b_all = w.tb_h0; a.midb_all = w.tb_mid; a.h1b_all = w.tb_h1;
        }
        return launch_coco_dec_fwd(a, s);
    }
    for (int t = 0; t < T; ++t) {
        const float* h0p = w.td_h0 + (size_t)t * RH; float* h0n = w.td_h0 + (size_t)(t + 1) * RH;
        const float* h1p = w.td_h1 + (size_t)t * RH; float* h1n = w.td_h1 + (size_t)(t + 1) * RH;
        const float* win = t == 0 ? sos : sentence + (size_t)(t - 1) * E;
        const long long ldwin = t == 0 ? 0 : (long long)T * E;
        GruLayer a{};
        a.x = win; a.ldx = ldwin; a.Kx = E; a.Wih = p + P.td0.wih; a.ldwih = in0; a.gi_add = w.td_zi0; a.ldgi = G;
        a.hprev = h0p; a.Whh = p + P.td0.whh; a.bhh = p + P.td0.bhh; a.h = h0n; a.rows = R;
        a.sav = save ? w.td_sav0 + (size_t)t * R * 4 * H : nullptr;
        const float* mid = h0n;
        if (keep) { a.keep = keep + (size_t)t * RH; a.scale = scale; a.mid = w.td_mid + (size_t)t * RH; mid = a.mid; }
        MMVAE_TRY(gru_layer_fwd(a, s));
        GruLayer b{};
        b.x = mid; b.ldx = H; b.Kx = H; b.Wih = p + P.td1.wih; b.ldwih = H; b.bih = p + P.td1.bih;
        b.hprev = h1p; b.Whh = p + P.td1.whh; b.bhh = p + P.td1.bhh; b.h = h1n; b.rows = R;
        b.sav = save ? w.td_sav1 + (size_t)t * R * 4 * H : nullptr;
        MMVAE_TRY(gru_layer_fwd(b, s));
        MMVAE_TRY(lin(h1n, H, R, p + P.td_h2o_w, E, H, ino, 0, nullptr, w.td_zo, E, sentence + (size_t)t * E, (long long)T * E, s));
    }
    return MMVAE_OK;
}

// bf16 persistent path: BPTT in one launch, then the weight gradients as batched bf16 GEMMs over all T*R rows
static int coco_text_dec_bwd_bf16(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, float* dw, float* dz,
                                  hipStream_t s, hipStream_t sw) {
    CocoPlan::W& w = P.w;
    const int R = groups * P.B, T = P.T, D = P.D, in0 = E + D, ino = H + D;
    const float* p = P.buf.params;
    const float scale = 1.f / (1.f - DROP_P);
    CocoDecBwdArgs a{};
    a.R = R; a.T = T; a.dw = dw; a.keep = keep; a.keep_scale = scale;
    auto pw = [&](int i) { return P.buf.packed + P.pk.d[i].dst_off; };
    a.w_hoT = pw(P.tb_hoT); a.w_ih1T = pw(P.tb_ih1T); a.w_hh1T = pw(P.tb_hh1T); a.w_hh0T = pw(P.tb_hh0T); a.w_ih0T = pw(P.tb_ih0T);
    a.h0_all = w.td_h0; a.h1_all = w.td_h1; a.sav0 = w.td_sav0; a.sav1 = w.td_sav1;
    a.dout_b = w.tb_dout; a.dgi0_b = w.tb_dgi0; a.dgh0_b = w.tb_dgh0; a.dgi1_b = w.tb_dgi1; a.dgh1_b = w.tb_dgh1;
    a.dhinit = w.td_dhinit; a.dwsum = w.td_dwsum;
    {
        const int Pc = coco_dec_cluster(R);
        if (Pc > 1) {
            a.cluster = Pc;
            a.cl_xchg = reinterpret_cast<unsigned long long*>(w.clb_xchg);
            a.cl_timeout = reinterpret_cast<unsigned*>(w.clb_xchg + w.clb_bytes - 64);
            P.cl_alarm_b = a.cl_timeout;
            MMVAE_TRY(launch_fill_zero(w.clb_xchg, w.clb_bytes, s));        // flags and the timeout word: zero before EVERY launch
            if (coco_dec_composed(P, Pc) && !mmvae_knob("coco_no_comb_bwd", 0)) {
                MMVAE_TRY(coco_text_dec_prepare(P, sos, s));                // (no-op inside a step: the forward pass made W_comb)
                if (!P.dw16_fresh) MMVAE_TRY(launch_coco_dw16(dw, (long long)R * T, w.tb_dw16, s));    // (normally made by the forward kernel with the fused MSE)
                a.w_combT = w.tb_combT; a.dw16 = w.tb_dw16; a.dzi0 = w.td_dzi0; a.dzi1 = w.td_dzi1;
            }
        }
    }
    MMVAE_TRY(launch_coco_dec_bwd(a, s));
    const bool composed = a.w_combT != nullptr;
    if (!composed) MMVAE_TRY(launch_coco_time_sum_bf16(a.dgi0_b, T, R, CTB_GP, G, w.td_dzi0, s));     // (the composed kernel keeps the sums itself)
    // composed form: the kernel left the time sum of dw in dwsum; the feedback part of sum_t dOut[t] is (sum_{t>=1} dgi0[t]) W_ih0x
    if (composed) MMVAE_TRY(lin_dx(w.td_dzi1, G, R, p + P.td0.wih, G, E, in0, 0, w.td_dwsum, E, 1, s));
    P.dec_wg_composed = composed; P.dec_wg_dw = dw;
    // ---- what the rest of the step waits for: dz through the three z-terms (initial state, layer-0 input, output projection)
    MMVAE_TRY(lin_dx(w.td_dhinit, H, R, p + P.td_z2h_w, H, D, D, 0, dz, D, 0, s));
    MMVAE_TRY(lin_dx(w.td_dzi0, G, R, p + P.td0.wih, G, D, in0, E, dz, D, 1, s));
    MMVAE_TRY(lin_dx(w.td_dwsum, E, R, p + P.td_h2o_w, E, D, ino, H, dz, D, 1, s));
    // ---- the weight gradients only feed the optimizer: the step issues them (coco_text_dec_wgrads) on the side stream
    // once the caption encoder's BPTT launch -- 8 workgroups, the rest of the chip idle -- is under way
    P.dec_wg_pending = true; P.dec_wg_z = z; P.dec_wg_groups = groups;
    if (sw == s) return coco_text_dec_wgrads(P, s);
    return MMVAE_OK;
}

// composed BPTT form: dOut[t] = dw[t] + dgi0[t+1] W_ih0x for every step at once -- one bf16 GEMM over the saved gate gradients
// (slices 1 .. T-1, fp32 result in a buffer of the fp32 path) + a pass that adds the loss term and lays dOut out as the
// operand of W_ho's weight gradient
static int coco_text_dec_dout(CocoPlan& P, const float* dw, int R, hipStream_t sw) {
    CocoPlan::W& w = P.w;
    const int T = P.T, M = (T - 1) * R;
    if (M > 0) {
        GatherPlan pl = dense_plan(M, G, CTB_GP, E);
        GemmParams q = gemm_of(P, pl, &P.tb_ih0xT_rm, 1, M);
        q.c.A = w.tb_dgi0 + (size_t)R * CTB_GP; q.out_f = w.td_dgi0; q.ldo = E;
        MMVAE_TRY(launch_gemm_gather(q, sw));
    }
    return launch_coco_dout_combine(dw, w.td_dgi0, T, R, w.tb_dout, sw);
}

// weight gradients of the bf16 caption decoder: dW[N][K] = P[T*R][N]^T G[T*R][K], bf16 operands saved in [t][row] layout.
// The caller has ordered `sw` behind the decoder's BPTT launch and its time sums.
int coco_text_dec_wgrads(CocoPlan& P, hipStream_t sw) {
    if (!P.dec_wg_pending) return MMVAE_OK;
    P.dec_wg_pending = false;
    CocoPlan::W& w = P.w;
    const float* z = P.dec_wg_z;
    const int R = P.dec_wg_groups * P.B, T = P.T, D = P.D, in0 = E + D, ino = H + D, TRn = T * R;
    float* g = P.buf.grads;
    WgradParams list[5];
    auto wg = [&](int i, int gidx, const bf16* Pm, int N, int ldp, const bf16* Gm, int C) {
        GatherPlan pl = dense_plan(TRn, C, C, N);
        WgradParams q = wgrad_of(P, pl, &gidx, 1, TRn);
        q.c.A = Gm; q.P = Pm; q.ldp = ldp;
        list[i] = q;
    };
    const size_t RHP = (size_t)R * CTB_HP;
    if (P.dec_wg_composed) MMVAE_TRY(coco_text_dec_dout(P, P.dec_wg_dw, R, sw));
    wg(0, P.tg_ih0, w.tb_dgi0, G, CTB_GP, w.tb_x, CTB_XP);
    wg(1, P.tg_hh0, w.tb_dgh0, G, CTB_GP, w.tb_h0, CTB_HP);                    // h0 BEFORE each step: slices 0 .. T-1
    wg(2, P.tg_ih1, w.tb_dgi1, G, CTB_GP, w.tb_mid, CTB_HP);
    wg(3, P.tg_hh1, w.tb_dgh1, G, CTB_GP, w.tb_h1, CTB_HP);
    wg(4, P.tg_ho, w.tb_dout, E, CTB_EP, w.tb_h1 + RHP, CTB_HP);               // h1 AFTER each step: slices 1 .. T
    MMVAE_TRY(launch_wgrad_group(list, 5, sw, &P.slab));
    MMVAE_TRY(launch_wgrad_reduce(&P.slab, sw, true));
    // (b_hh0, b_ih1, b_hh1: column 200 of the three hidden-state operands is 1.0, their gradients are column 200 above)
    // initial hidden state h = z2h(z), shared by both layers; time-invariant z terms (as in the fp32 path)
    MMVAE_TRY(lin_dw(w.td_dhinit, H, z, D, R, g + P.td_z2h_w, H, D, D, 0, sw));
    MMVAE_TRY(launch_colsum_f32(w.td_dhinit, R, H, g + P.td_z2h_b, sw));
    MMVAE_TRY(lin_dw(w.td_dzi0, G, z, D, R, g + P.td0.wih, G, D, in0, E, sw));
    MMVAE_TRY(launch_colsum_f32(w.td_dzi0, R, G, g + P.td0.bih, sw));
    MMVAE_TRY(lin_dw(w.td_dwsum, E, z, D, R, g + P.td_h2o_w, E, D, ino, H, sw));
    return launch_colsum_f32(w.td_dwsum, R, E, g + P.td_h2o_b, sw);
}

int coco_text_dec_bwd(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, const float* sentence, float* dw,
                      float* dz, hipStream_t s, hipStream_t sw, bool bf16_path) {
    if (bf16_path && P.text_bf16) return coco_text_dec_bwd_bf16(P, z, groups, sos, keep, dw, dz, s, sw);
    CocoPlan::W& w = P.w;
    const int R = groups * P.B, T = P.T, D = P.D, in0 = E + D, ino = H + D;
    const float* p = P.buf.params;
    float* g = P.buf.grads;
    const size_t RH = (size_t)R * H, RG = (size_t)R * G;
    const long long ldw = (long long)T * E;
    const float scale = 1.f / (1.f - DROP_P);
    // td_dh0 / td_dh1 (carried gradients of the two hidden states) start at zero: the caller zeroes them.
    // Weight gradients only feed the optimizer: they go to `sw` (a side stream, or `s` itself) behind an event edge, off
    // the 102-step dependency chain.  Their operands are never rewritten after the edge (per-step buffers; dw[:, t] is
    // final once step t starts).
    const bool fork = sw != s;
    auto to_side = [&]() -> int { return fork ? edge(P, s, sw) : MMVAE_OK; };
    for (int t = T - 1; t >= 0; --t) {
        const float* h0p = w.td_h0 + (size_t)t * RH; const float* h1p = w.td_h1 + (size_t)t * RH;
        const float* h1n = w.td_h1 + (size_t)(t + 1) * RH;
        const float* mid = keep ? w.td_mid + (size_t)t * RH : w.td_h0 + (size_t)(t + 1) * RH;
        float* dwt = dw + (size_t)t * E;                 // total gradient wrt this step's output (loss + feedback of step t+1)
        float *dgi1 = w.td_dgi1 + (size_t)t * RG, *dgh1 = w.td_dgh1 + (size_t)t * RG;
        float *dgi0 = w.td_dgi0 + (size_t)t * RG, *dgh0 = w.td_dgh0 + (size_t)t * RG;
        // output projection
        MMVAE_TRY(lin_dx(dwt, ldw, R, p + P.td_h2o_w, E, H, ino, 0, w.td_dh1, H, 1, s));
        MMVAE_TRY(to_side());
        MMVAE_TRY(lin_dw(dwt, ldw, h1n, H, R, g + P.td_h2o_w, E, H, ino, 0, sw));
        // layer 1
        GruBwd b{};
        b.dh = w.td_dh1; b.sav = w.td_sav1 + (size_t)t * R * 4 * H; b.hprev = h1p; b.dgi = dgi1; b.lddgi = G; b.dgh = dgh1;
        b.dh_out = w.td_dh1; b.rows = R;
        MMVAE_TRY(gru_bwd(b, s));
        MMVAE_TRY(gemm_f32_pair(mk_dx(dgi1, G, R, p + P.td1.wih, G, H, H, 0, w.td_dmid, H, 0),
                                mk_dx(dgh1, G, R, p + P.td1.whh, G, H, H, 0, w.td_dh1, H, 1), s));
        // layer 0 (its output fed layer 1 through the dropout)
        GruBwd a{};
        a.dh = w.td_dh0; a.dh2 = w.td_dmid; a.keep = keep ? keep + (size_t)t * RH : nullptr; a.scale = scale;
        a.sav = w.td_sav0 + (size_t)t * R * 4 * H; a.hprev = h0p; a.dgi = dgi0; a.lddgi = G; a.dgh = dgh0; a.dh_out = w.td_dh0; a.rows = R;
        MMVAE_TRY(gru_bwd(a, s));
        // hidden-state path, and the input vector of this step: the previous output (gradient flows back into it) or
        // '<s>' (a constant)
        if (t > 0)
            MMVAE_TRY(gemm_f32_pair(mk_dx(dgh0, G, R, p + P.td0.whh, G, H, H, 0, w.td_dh0, H, 1),
                                    mk_dx(dgi0, G, R, p + P.td0.wih, G, E, in0, 0, dw + (size_t)(t - 1) * E, ldw, 1), s));
        else
            MMVAE_TRY(lin_dx(dgh0, G, R, p + P.td0.whh, G, H, H, 0, w.td_dh0, H, 1, s));
        const float* win = t == 0 ? sos : sentence + (size_t)(t - 1) * E;
        MMVAE_TRY(to_side());
        MMVAE_TRY(lin_dw(dgi0, G, win, t == 0 ? 0 : ldw, R, g + P.td0.wih, G, E, in0, 0, sw));
        (void)mid;
    }
    const int TR = T * R;
    // weight gradients whose operand pairs share the [t][row] layout: one GEMM over K = T*R each
    MMVAE_TRY(to_side());
    MMVAE_TRY(lin_dw(w.td_dgh0, G, w.td_h0, H, TR, g + P.td0.whh, G, H, H, 0, sw));
    MMVAE_TRY(lin_dw(w.td_dgh1, G, w.td_h1, H, TR, g + P.td1.whh, G, H, H, 0, sw));
    MMVAE_TRY(lin_dw(w.td_dgi1, G, keep ? w.td_mid : w.td_h0 + RH, H, TR, g + P.td1.wih, G, H, H, 0, sw));
    MMVAE_TRY(launch_colsum_f32(w.td_dgh0, TR, G, g + P.td0.bhh, sw));
    MMVAE_TRY(launch_colsum_f32(w.td_dgi1, TR, G, g + P.td1.bih, sw));
    MMVAE_TRY(launch_colsum_f32(w.td_dgh1, TR, G, g + P.td1.bhh, sw));
    // initial hidden state h = z2h(z), shared by both layers
    MMVAE_TRY(add2(w.td_dh0, w.td_dh1, (long long)RH, w.td_dhinit, s));
    MMVAE_TRY(lin_dw(w.td_dhinit, H, z, D, R, g + P.td_z2h_w, H, D, D, 0, s));
    MMVAE_TRY(launch_colsum_f32(w.td_dhinit, R, H, g + P.td_z2h_b, s));
    MMVAE_TRY(lin_dx(w.td_dhinit, H, R, p + P.td_z2h_w, H, D, D, 0, dz, D, 0, s));
    // time-invariant z terms: gradients of the time-summed pre-activations
    MMVAE_TRY(sum_mid(w.td_dgi0, 1, T, (long long)RG, w.td_dzi0, s));
    MMVAE_TRY(lin_dw(w.td_dzi0, G, z, D, R, g + P.td0.wih, G, D, in0, E, s));
    MMVAE_TRY(launch_colsum_f32(w.td_dzi0, R, G, g + P.td0.bih, s));
    MMVAE_TRY(lin_dx(w.td_dzi0, G, R, p + P.td0.wih, G, D, in0, E, dz, D, 1, s));
    MMVAE_TRY(sum_mid(dw, R, T, E, w.td_dwsum, s));
    MMVAE_TRY(lin_dw(w.td_dwsum, E, z, D, R, g + P.td_h2o_w, E, D, ino, H, s));
    MMVAE_TRY(launch_colsum_f32(w.td_dwsum, R, E, g + P.td_h2o_b, s));
    return lin_dx(w.td_dwsum, E, R, p + P.td_h2o_w, E, D, ino, H, dz, D, 1, s);
}

int coco_mse3(const float* recon, const float* target, int Gn, long long per_group, const float* coef, float* loss_sum, float* dw,
              hipStream_t s) {
    MMVAE_REQUIRE(Gn >= 1 && Gn <= 3, "mse3: groups=%d", Gn);
    const unsigned nb = (unsigned)std::min<long long>((per_group + TPBT - 1) / TPBT, 4096);
    hipLaunchKernelGGL(mse3_kernel, dim3(nb, Gn), dim3(TPBT), 0, s, recon, target, per_group, coef[0], coef[1], coef[2], loss_sum, dw);
    return mmvae_check_launch("mse3");
}
