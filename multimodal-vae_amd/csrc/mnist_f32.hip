// fp32 path of the MNIST MMVAE (mnist/model.py:14-185, mnist/train.py:64-81,131-147).
//
// Why this model gets its own precision.  It is Linear -> BatchNorm1d -> ReLU: ReLU makes the gradient discontinuous in
// the operand rounding of a bf16 MFMA path (an fp32 run of the reference with ONLY its GEMM operands rounded to bf16
// moves image_encoder.net.0.weight.grad by 12-32 %, tools/sim_bf16_mnist.py), and the whole step is 1.4 GFLOP, i.e.
// launch-latency sized on this chip at any precision.  So the default MNIST plan computes exactly what the reference
// computes: fp32 operands, fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate), fp32 activations,
// two-pass BatchNorm statistics.  Parity against the reference is then at fp32 level (tests/test_gpu_mnist.py).
//
// Kernels: one generic strided GEMM (forward / data gradient / weight gradient are the same kernel with different
// strides -- the fp32 MFMA operand fragment is ONE float per lane, so any stride pattern loads coalesced), BatchNorm1d
// forward (+ReLU) and backward as one thread per channel walking its group's rows, embedding gather / scatter.
// Losses, product of experts, reparametrisation, KL and Adam are the shared fp32 kernels of elementwise.hip.
#include "mnist_plan.h"
#include "gemm_f32.h"

namespace {

constexpr int TPBF = 256;

// y[rows][N] = x[rows][K] W[N][K]^T + b
int lin_fwd32(MnistPlan& P, const MlpLin& L, const float* x, int ldx, int rows, float* y, hipStream_t s) {
    F32Gemm g{};
    g.A = x; g.a_rs = ldx; g.a_cs = 1; g.B = P.buf.params + L.w_off; g.b_rs = 1; g.b_cs = L.K;
    g.M = rows; g.N = L.N; g.K = L.K; g.C = y; g.ldc = L.N; g.bias = P.buf.params + L.b_off;
    return gemm_f32(g, s);
}
// dx[rows][K] = dy[rows][N] W[N][K]
int lin_dgrad32(MnistPlan& P, const MlpLin& L, const float* dy, int rows, float* dx, hipStream_t s) {
    F32Gemm g{};
    g.A = dy; g.a_rs = L.N; g.a_cs = 1; g.B = P.buf.params + L.w_off; g.b_rs = L.K; g.b_cs = 1;
    g.M = rows; g.N = L.K; g.K = L.N; g.C = dx; g.ldc = L.K;
    return gemm_f32(g, s);
}
// dW[N][K] = dy[rows][N]^T x[rows][K]   (the step zeroes grads first; every weight has exactly one writer: plain store)
int lin_wgrad32(MnistPlan& P, const MlpLin& L, const float* dy, const float* x, int ldx, int rows, hipStream_t s) {
    F32Gemm g{};
    g.A = dy; g.a_rs = 1; g.a_cs = L.N; g.B = x; g.b_rs = ldx; g.b_cs = 1;
    g.M = L.N; g.N = L.K; g.K = rows; g.C = P.buf.grads + L.w_off; g.ldc = L.K;
    return gemm_f32(g, s);
}

// ------------------------------------------------------------------ BatchNorm1d (+ ReLU), fp32, one thread per (group, channel)
struct Bn32 {
    const float* x; float* y;            // [G*rpg][C] raw in / activated out
    int C, G, rpg;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; long long* nbt;
    float2* mr;                          // [G][C] (mean, rstd) out
    int updates; unsigned skip_mask; int training; int relu;
};
// workgroup = 64 channels x 16 row lanes (1024 threads): every pass over a group's rows is rpg/16 coalesced steps
constexpr int BN_RL = 16;
__device__ __forceinline__ float bn_block_sum(float v, float (*sh)[64], int cl, int rl) {
    __syncthreads();
    sh[rl][cl] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < BN_RL; ++q) s += sh[q][cl];
    return s;
}
__global__ __launch_bounds__(64 * BN_RL) void bn1d_fwd32_kernel(const Bn32 a) {
    __shared__ float sh[BN_RL][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < a.C;
    const int cc = cok ? c : 0;
    const float gamma = a.gamma[cc], beta = a.beta[cc];
    float rm = a.running_mean[cc], rv = a.running_var[cc];
    for (int g = 0; g < a.G; ++g) {
        const float* x = a.x + (size_t)g * a.rpg * a.C + cc;
        float* y = a.y + (size_t)g * a.rpg * a.C + cc;
        float mean, rstd;
        if (a.training) {
            float s = 0.f;
            for (int r = rl; r < a.rpg; r += BN_RL) s += x[(size_t)r * a.C];
            mean = bn_block_sum(s, sh, cl, rl) / (float)a.rpg;
            float v = 0.f;
            for (int r = rl; r < a.rpg; r += BN_RL) { const float d = x[(size_t)r * a.C] - mean; v += d * d; }
            const float var = bn_block_sum(v, sh, cl, rl) / (float)a.rpg;       // biased, like nn.BatchNorm1d's normalisation
            rstd = rsqrtf(var + BN_EPS);
            const int nu = ((a.skip_mask >> g) & 1u) ? 0 : a.updates;
            const float unbiased = var * (float)a.rpg / (float)(a.rpg - 1);
            for (int u = 0; u < nu; ++u) {
                rm = (1.f - BN_MOM) * rm + BN_MOM * mean;
                rv = (1.f - BN_MOM) * rv + BN_MOM * unbiased;
            }
        } else {
            mean = rm; rstd = rsqrtf(rv + BN_EPS);
        }
        if (cok && rl == 0) a.mr[g * a.C + c] = make_float2(mean, rstd);
        if (cok)
            for (int r = rl; r < a.rpg; r += BN_RL) {
                float o = (x[(size_t)r * a.C] - mean) * rstd * gamma + beta;
                if (a.relu) o = fmaxf(o, 0.f);
                y[(size_t)r * a.C] = o;
            }
    }
    if (a.training && cok && rl == 0) {
        a.running_mean[c] = rm; a.running_var[c] = rv;
        if (c == 0) *a.nbt += (long long)(a.G - __popc(a.skip_mask & ((1u << a.G) - 1u))) * a.updates;
    }
}
int bn_fwd32(MnistPlan& P, int bi, const float* x, float* y, int rows, int groups, int updates, int training, float2* mr, hipStream_t s) {
    const BnL& b = P.bn[bi];
    MMVAE_REQUIRE(!training || rows / groups > 1, "Expected more than 1 value per channel when training");
    Bn32 a{};
    a.x = x; a.y = y; a.C = b.C; a.G = groups; a.rpg = rows / groups;
    a.gamma = P.buf.params + b.w_off; a.beta = P.buf.params + b.b_off;
    a.running_mean = P.buf.bn_stats + b.stat_off; a.running_var = P.buf.bn_stats + b.stat_off + b.C; a.nbt = P.buf.bn_nbt + b.idx;
    a.mr = mr; a.updates = updates; a.skip_mask = groups > 1 ? P.dec_skip_mask : 0u; a.training = training; a.relu = 1;
    hipLaunchKernelGGL(bn1d_fwd32_kernel, dim3(ceil_div(b.C, 64)), dim3(64 * BN_RL), 0, s, a);
    return mmvae_check_launch("bn1d_fwd32");
}
// d: grad wrt the activated output (in), grad wrt the raw input (out, in place)
struct BnB32 {
    float* d; const float* x; int C, G, rpg;
    const float* gamma; const float* beta; const float2* mr;
    float* dgamma; float* dbeta; int training;
};
__global__ __launch_bounds__(64 * BN_RL) void bn1d_bwd32_kernel(const BnB32 a) {
    __shared__ float sh[BN_RL][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < a.C;
    const int cc = cok ? c : 0;
    const float gamma = a.gamma[cc], beta = a.beta[cc];
    float dg = 0.f, db = 0.f;
    for (int g = 0; g < a.G; ++g) {
        const float2 m = a.mr[g * a.C + cc];
        const float* x = a.x + (size_t)g * a.rpg * a.C + cc;
        float* d = a.d + (size_t)g * a.rpg * a.C + cc;
        float p1 = 0.f, p2 = 0.f;
        for (int r = rl; r < a.rpg; r += BN_RL) {
            const float xh = (x[(size_t)r * a.C] - m.x) * m.y;
            const float dy = (xh * gamma + beta > 0.f) ? d[(size_t)r * a.C] : 0.f;     // ReLU backward
            p1 += dy; p2 += dy * xh;
        }
        const float s1 = bn_block_sum(p1, sh, cl, rl);
        const float s2 = bn_block_sum(p2, sh, cl, rl);
        dg += s2; db += s1;
        const float inv = 1.f / (float)a.rpg;
        if (cok)
            for (int r = rl; r < a.rpg; r += BN_RL) {
                const float xh = (x[(size_t)r * a.C] - m.x) * m.y;
                const float dy = (xh * gamma + beta > 0.f) ? d[(size_t)r * a.C] : 0.f;
                d[(size_t)r * a.C] = a.training ? gamma * m.y * (dy - s1 * inv - xh * s2 * inv) : gamma * m.y * dy;
            }
    }
    if (cok && rl == 0) { a.dgamma[c] += dg; a.dbeta[c] += db; }
}
int bn_bwd32(MnistPlan& P, int bi, float* d, const float* x, int rows, int groups, const float2* mr, hipStream_t s) {
    const BnL& b = P.bn[bi];
    BnB32 a{};
    a.d = d; a.x = x; a.C = b.C; a.G = groups; a.rpg = rows / groups;
    a.gamma = P.buf.params + b.w_off; a.beta = P.buf.params + b.b_off; a.mr = mr;
    a.dgamma = P.buf.grads + b.w_off; a.dbeta = P.buf.grads + b.b_off; a.training = 1;
    hipLaunchKernelGGL(bn1d_bwd32_kernel, dim3(ceil_div(b.C, 64)), dim3(64 * BN_RL), 0, s, a);
    return mmvae_check_launch("bn1d_bwd32");
}

// ------------------------------------------------------------------ label embedding (mnist/model.py:141)
__global__ void embed_gather32_kernel(const float* table, int C, const long long* idx, int rows, float* x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const int r = i / C, c = i - r * C;
    x[i] = table[idx[r] * C + c];
}
__global__ void embed_scatter32_kernel(const float* d, int C, const long long* idx, int rows, float* g_table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const int r = i / C, c = i - r * C;
    atomicAdd(g_table + idx[r] * C + c, d[i]);
}

void carve32(MnistPlan& P, Workspace& ws) {
    MnistPlan::W32& w = P.w32;
    const size_t B = P.B, D = P.D, B3 = 3 * B;
    const int bnc[6] = {400, 200, 200, 400, 50, 10};
    char* z0 = ws.take<char>(0);
    w.sums = ws.take<float>(16 * MMVAE_LOSS_SLOTS);
    w.dz_img = ws.take<float>(B3 * D); w.dz_txt = ws.take<float>(B3 * D);
    char* z1 = ws.take<char>(0);
    w.zero_begin = z0; w.zero_bytes = (size_t)(z1 - z0);
    for (int i = 0; i < 6; ++i) w.mr[i] = ws.take<float2>(3 * bnc[i]);
    w.r_ie[0] = ws.take<float>(B * 400); w.a_ie[0] = ws.take<float>(B * 400);
    w.r_ie[1] = ws.take<float>(B * 200); w.a_ie[1] = ws.take<float>(B * 200);
    w.encout = ws.take<float>(B * 2 * D);
    w.r_te = ws.take<float>(B * 50); w.a_te = ws.take<float>(B * 50); w.txtout = ws.take<float>(B * 2 * D);
    w.eps = ws.take<float>(B3 * D); w.mu = ws.take<float>(B3 * D); w.logvar = ws.take<float>(B3 * D);
    w.z = ws.take<float>(B3 * D); w.z_bf = ws.take<bf16>(B3 * P.ldz);
    w.r_id[0] = ws.take<float>(B3 * 200); w.a_id[0] = ws.take<float>(B3 * 200);
    w.r_id[1] = ws.take<float>(B3 * 400); w.a_id[1] = ws.take<float>(B3 * 400);
    w.logits = ws.take<float>(B3 * 784); w.dlogit = ws.take<float>(B3 * 784);
    w.r_td = ws.take<float>(B3 * 10); w.a_td = ws.take<float>(B3 * 10);
    w.tlogits = ws.take<float>(B3 * 10); w.words = ws.take<float>(B3 * 10); w.dtl = ws.take<float>(B3 * 10);
    w.d_id[0] = ws.take<float>(B3 * 200); w.d_id[1] = ws.take<float>(B3 * 400); w.d_td = ws.take<float>(B3 * 10);
    w.d_encout = ws.take<float>(B * 2 * D); w.d_txtout = ws.take<float>(B * 2 * D);
    w.d_ie[0] = ws.take<float>(B * 400); w.d_ie[1] = ws.take<float>(B * 200); w.d_te = ws.take<float>(B * 50);
}

int use_ws32(MnistPlan& P, void* ws, size_t bytes) {
    MMVAE_REQUIRE(ws != nullptr && bytes >= P.ws_bytes, "workspace too small (%zu < %zu)", bytes, P.ws_bytes);
    Workspace w(ws, bytes);
    carve32(P, w);
    return MMVAE_OK;
}

// ---- module pieces (rows = B for the encoders, groups*B for the decoders)
int img_enc_fwd32(MnistPlan& P, const float* image, int updates, int training, float* out, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int B = P.B;
    MMVAE_TRY(lin_fwd32(P, P.ie[0], image, 784, B, w.r_ie[0], s));
    MMVAE_TRY(bn_fwd32(P, 0, w.r_ie[0], w.a_ie[0], B, 1, updates, training, w.mr[0], s));
    MMVAE_TRY(lin_fwd32(P, P.ie[1], w.a_ie[0], 400, B, w.r_ie[1], s));
    MMVAE_TRY(bn_fwd32(P, 1, w.r_ie[1], w.a_ie[1], B, 1, updates, training, w.mr[1], s));
    return lin_fwd32(P, P.ie[2], w.a_ie[1], 200, B, out, s);
}
int img_enc_bwd32(MnistPlan& P, const float* image, const float* d_out, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int B = P.B;
    MMVAE_TRY(launch_colsum_f32(d_out, B, 2 * P.D, P.buf.grads + P.ie[2].b_off, s));
    MMVAE_TRY(lin_wgrad32(P, P.ie[2], d_out, w.a_ie[1], 200, B, s));
    MMVAE_TRY(lin_dgrad32(P, P.ie[2], d_out, B, w.d_ie[1], s));
    MMVAE_TRY(bn_bwd32(P, 1, w.d_ie[1], w.r_ie[1], B, 1, w.mr[1], s));
    MMVAE_TRY(lin_wgrad32(P, P.ie[1], w.d_ie[1], w.a_ie[0], 400, B, s));
    MMVAE_TRY(lin_dgrad32(P, P.ie[1], w.d_ie[1], B, w.d_ie[0], s));
    MMVAE_TRY(bn_bwd32(P, 0, w.d_ie[0], w.r_ie[0], B, 1, w.mr[0], s));
    return lin_wgrad32(P, P.ie[0], w.d_ie[0], image, 784, B, s);
    // (biases in front of a BatchNorm have an exactly zero gradient: left at the zero the step wrote)
}
int txt_enc_fwd32(MnistPlan& P, const long long* label, int updates, int training, float* out, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int B = P.B;
    hipLaunchKernelGGL(embed_gather32_kernel, dim3(ceil_div(B * 50, 256)), dim3(256), 0, s, P.buf.params + P.emb_off, 50, label, B, w.r_te);
    MMVAE_TRY(mmvae_check_launch("embed_gather32"));
    MMVAE_TRY(bn_fwd32(P, 4, w.r_te, w.a_te, B, 1, updates, training, w.mr[4], s));
    return lin_fwd32(P, P.te_lin, w.a_te, 50, B, out, s);
}
int txt_enc_bwd32(MnistPlan& P, const long long* label, const float* d_out, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int B = P.B;
    MMVAE_TRY(launch_colsum_f32(d_out, B, 2 * P.D, P.buf.grads + P.te_lin.b_off, s));
    MMVAE_TRY(lin_wgrad32(P, P.te_lin, d_out, w.a_te, 50, B, s));
    MMVAE_TRY(lin_dgrad32(P, P.te_lin, d_out, B, w.d_te, s));
    MMVAE_TRY(bn_bwd32(P, 4, w.d_te, w.r_te, B, 1, w.mr[4], s));
    hipLaunchKernelGGL(embed_scatter32_kernel, dim3(ceil_div(B * 50, 256)), dim3(256), 0, s, w.d_te, 50, label, B, P.buf.grads + P.emb_off);
    return mmvae_check_launch("embed_scatter32");
}
int img_dec_fwd32(MnistPlan& P, const float* z, int groups, int training, float* logits, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int rows = groups * P.B;
    MMVAE_TRY(lin_fwd32(P, P.id[0], z, P.D, rows, w.r_id[0], s));
    MMVAE_TRY(bn_fwd32(P, 2, w.r_id[0], w.a_id[0], rows, groups, 1, training, w.mr[2], s));
    MMVAE_TRY(lin_fwd32(P, P.id[1], w.a_id[0], 200, rows, w.r_id[1], s));
    MMVAE_TRY(bn_fwd32(P, 3, w.r_id[1], w.a_id[1], rows, groups, 1, training, w.mr[3], s));
    return lin_fwd32(P, P.id[2], w.a_id[1], 400, rows, logits, s);
}
int img_dec_bwd32(MnistPlan& P, const float* z, const float* dlogit, int groups, float* dz, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int rows = groups * P.B;
    MMVAE_TRY(launch_colsum_f32(dlogit, rows, 784, P.buf.grads + P.id[2].b_off, s));
    MMVAE_TRY(lin_wgrad32(P, P.id[2], dlogit, w.a_id[1], 400, rows, s));
    MMVAE_TRY(lin_dgrad32(P, P.id[2], dlogit, rows, w.d_id[1], s));
    MMVAE_TRY(bn_bwd32(P, 3, w.d_id[1], w.r_id[1], rows, groups, w.mr[3], s));
    MMVAE_TRY(lin_wgrad32(P, P.id[1], w.d_id[1], w.a_id[0], 200, rows, s));
    MMVAE_TRY(lin_dgrad32(P, P.id[1], w.d_id[1], rows, w.d_id[0], s));
    MMVAE_TRY(bn_bwd32(P, 2, w.d_id[0], w.r_id[0], rows, groups, w.mr[2], s));
    MMVAE_TRY(lin_wgrad32(P, P.id[0], w.d_id[0], z, P.D, rows, s));
    return lin_dgrad32(P, P.id[0], w.d_id[0], rows, dz, s);
}
int txt_dec_fwd32(MnistPlan& P, const float* z, int groups, int training, float* tlogits, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int rows = groups * P.B;
    MMVAE_TRY(lin_fwd32(P, P.td[0], z, P.D, rows, w.r_td, s));
    MMVAE_TRY(bn_fwd32(P, 5, w.r_td, w.a_td, rows, groups, 1, training, w.mr[5], s));
    return lin_fwd32(P, P.td[1], w.a_td, 10, rows, tlogits, s);
}
int txt_dec_bwd32(MnistPlan& P, const float* z, const float* dtl, int groups, float* dz, hipStream_t s) {
    MnistPlan::W32& w = P.w32; const int rows = groups * P.B;
    MMVAE_TRY(launch_colsum_f32(dtl, rows, 10, P.buf.grads + P.td[1].b_off, s));
    MMVAE_TRY(lin_wgrad32(P, P.td[1], dtl, w.a_td, 10, rows, s));
    MMVAE_TRY(lin_dgrad32(P, P.td[1], dtl, rows, w.d_td, s));
    MMVAE_TRY(bn_bwd32(P, 5, w.d_td, w.r_td, rows, groups, w.mr[5], s));
    MMVAE_TRY(lin_wgrad32(P, P.td[0], w.d_td, z, P.D, rows, s));
    return lin_dgrad32(P, P.td[0], w.d_td, rows, dz, s);
}
// dlogits[r][c] = d_logp[r][c] - softmax[r][c] * sum_c d_logp[r][c]   (log_softmax backward)
__global__ void logsoftmax_bwd32_kernel(const float* d_logp, const float* logp, int rows, int classes, float* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float sum = 0.f;
    for (int c = 0; c < classes; ++c) sum += d_logp[(size_t)r * classes + c];
    for (int c = 0; c < classes; ++c)
        out[(size_t)r * classes + c] = d_logp[(size_t)r * classes + c] - expf(logp[(size_t)r * classes + c]) * sum;
}

}  // namespace

size_t mnist_f32_workspace_bytes(MnistPlan& P) {
    Workspace ws(nullptr, 0);
    carve32(P, ws);
    return ws.used();
}

int mnist_f32_step(MnistPlan& P, const MnistStepIO& io, int training, int do_backward, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, io.ws, io.ws_bytes));
    MMVAE_REQUIRE(io.image && io.label && io.sums, "mnist step: image/label/sums must be given");
    MnistPlan::W32& w = P.w32;
    const int B = P.B, D = P.D, B3 = 3 * B;
    const float* eps = io.eps;
    StepBeginArgs sb{};
    sb.zero_ptr[0] = w.zero_begin; sb.zero_bytes[0] = w.zero_bytes;
    if (do_backward) { sb.zero_ptr[1] = P.buf.grads; sb.zero_bytes[1] = (size_t)(P.nparams / 4) * 16; }
    sb.seed = io.seed; sb.step = io.step_ctr;
    if (training && !eps) { sb.eps = w.eps; sb.n_eps = (long long)B3 * D; eps = w.eps; }
    MMVAE_TRY(launch_step_begin(sb, s));
    if (do_backward && P.nparams % 4 != 0)
        MMVAE_TRY(launch_fill_zero(P.buf.grads + (P.nparams / 4) * 4, (size_t)(P.nparams % 4) * sizeof(float), s));
    const int sk[3] = {io.pass_skip[0] != 0, io.pass_skip[1] != 0, io.pass_skip[2] != 0};
    P.dec_skip_mask = (unsigned)(sk[0] | (sk[1] << 1) | (sk[2] << 2));
    // ---- encoders, once each for the two passes that share their input (module docstring of mnist.hip)
    MMVAE_TRY(img_enc_fwd32(P, io.image, 2 - sk[0] - sk[1], training, w.encout, s));
    MMVAE_TRY(txt_enc_fwd32(P, io.label, 2 - sk[0] - sk[2], training, w.txtout, s));
    Latent3Args la{};
    la.B = B; la.D = D; la.img_out = w.encout; la.img_out_b = w.encout; la.txt_out = w.txtout; la.eps = eps;
    la.mu = io.mu ? io.mu : w.mu; la.logvar = io.logvar ? io.logvar : w.logvar;
    la.z_f32 = w.z; la.z_bf = w.z_bf; la.ldz = P.ldz; la.kl_sum = w.sums + 8; la.training = training;
    MMVAE_TRY(launch_latent3_fwd(la, s));
    // ---- decoders on 3B rows, BatchNorm per pass
    MMVAE_TRY(img_dec_fwd32(P, w.z, 3, training, w.logits, s));
    BceArgs bc{};
    bc.logits = w.logits; bc.ldl = 1; bc.target = io.image; bc.G = 3; bc.B = B; bc.C = 1; bc.H = 28; bc.W = 28;
    bc.recon = io.recon_image; bc.dlogit = do_backward ? w.dlogit : nullptr; bc.loss_sum = w.sums;
    for (int k = 0; k < 3; ++k) bc.coef[k] = sk[k] ? 0.f : io.lambda_xy[k] / (float)(B * 784);
    MMVAE_TRY(launch_sigmoid_bce(bc, s));
    MMVAE_TRY(txt_dec_fwd32(P, w.z, 3, training, w.tlogits, s));
    LogSoftmaxNllArgs ls{};
    ls.logits = w.tlogits; ls.rows = B3; ls.classes = 10; ls.words = io.recon_text ? io.recon_text : w.words;
    ls.target = io.label; ls.target_rows = B; ls.rows_per_group = B; ls.nll_sum = w.sums + 4;
    ls.dlogits_f32 = do_backward ? w.dtl : nullptr;
    for (int k = 0; k < 3; ++k) ls.coef[k] = sk[k] ? 0.f : io.lambda_yx[k] / (float)B;
    MMVAE_TRY(launch_logsoftmax_nll(ls, s));
    hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums);
    MMVAE_TRY(mmvae_check_launch("sum_slots"));
    if (!do_backward) return MMVAE_OK;
    // =============================== backward ===============================
    MMVAE_TRY(txt_dec_bwd32(P, w.z, w.dtl, 3, w.dz_txt, s));
    MMVAE_TRY(img_dec_bwd32(P, w.z, w.dlogit, 3, w.dz_img, s));
    Latent3BwdArgs lb{};
    lb.f = la; lb.dz_a = w.dz_img; lb.dz_b = w.dz_txt;
    for (int k = 0; k < 3; ++k) lb.kl_coef[k] = sk[k] ? 0.f : io.kl_coef;
    lb.d_img_out_f32 = w.d_encout; lb.d_txt_out = w.d_txtout;
    MMVAE_TRY(launch_latent3_bwd(lb, s));
    MMVAE_TRY(img_enc_bwd32(P, io.image, w.d_encout, s));
    return txt_enc_bwd32(P, io.label, w.d_txtout, s);
}

// ---------------------------------------------------------------- granular modules (B rows, one BatchNorm group)
int mnist_f32_image_encoder_fwd(MnistPlan& P, void* ws, size_t wsb, const float* image, int training, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    P.dec_skip_mask = 0;
    // the backward needs the input again: keep a copy in the call's workspace (d_ie[0] is free until then... no: use logits)
    MMVAE_TRY(launch_fill_zero(P.w32.zero_begin, P.w32.zero_bytes, s));
    if (hipMemcpyAsync(P.w32.logits, image, (size_t)P.B * 784 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        mmvae_set_error("hipMemcpyAsync failed"); return MMVAE_EHIP;
    }
    return img_enc_fwd32(P, P.w32.logits, 1, training, out, s);
}
int mnist_f32_image_encoder_bwd(MnistPlan& P, void* ws, size_t wsb, const float* d_out, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    return img_enc_bwd32(P, P.w32.logits, d_out, s);
}
int mnist_f32_image_decoder_fwd(MnistPlan& P, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    P.dec_skip_mask = 0;
    MnistPlan::W32& w = P.w32;
    if (hipMemcpyAsync(w.z, z, (size_t)P.B * P.D * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        mmvae_set_error("hipMemcpyAsync failed"); return MMVAE_EHIP;
    }
    MMVAE_TRY(img_dec_fwd32(P, w.z, 1, training, w.logits, s));
    const long long n = (long long)P.B * 784;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, w.logits, n, recon);
    return mmvae_check_launch("sigmoid");
}
int mnist_f32_image_decoder_bwd(MnistPlan& P, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    MnistPlan::W32& w = P.w32;
    const long long n = (long long)P.B * 784;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, d_recon, recon, n, w.dlogit);
    MMVAE_TRY(mmvae_check_launch("sigmoid_bwd"));
    return img_dec_bwd32(P, w.z, w.dlogit, 1, dz, s);
}
int mnist_f32_text_encoder_fwd(MnistPlan& P, void* ws, size_t wsb, const long long* label, int training, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    P.dec_skip_mask = 0;
    return txt_enc_fwd32(P, label, 1, training, out, s);
}
int mnist_f32_text_encoder_bwd(MnistPlan& P, void* ws, size_t wsb, const long long* label, const float* d_out, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    return txt_enc_bwd32(P, label, d_out, s);
}
int mnist_f32_text_decoder_fwd(MnistPlan& P, void* ws, size_t wsb, const float* z, int training, float* logp, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    P.dec_skip_mask = 0;
    MnistPlan::W32& w = P.w32;
    if (hipMemcpyAsync(w.z, z, (size_t)P.B * P.D * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        mmvae_set_error("hipMemcpyAsync failed"); return MMVAE_EHIP;
    }
    MMVAE_TRY(txt_dec_fwd32(P, w.z, 1, training, w.tlogits, s));
    LogSoftmaxNllArgs ls{};
    ls.logits = w.tlogits; ls.rows = P.B; ls.classes = 10; ls.words = logp; ls.rows_per_group = P.B; ls.target_rows = P.B;
    return launch_logsoftmax_nll(ls, s);
}
int mnist_f32_text_decoder_bwd(MnistPlan& P, void* ws, size_t wsb, const float* d_logp, const float* logp, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws32(P, ws, wsb));
    MnistPlan::W32& w = P.w32;
    hipLaunchKernelGGL(logsoftmax_bwd32_kernel, dim3(ceil_div(P.B, 256)), dim3(256), 0, s, d_logp, logp, P.B, 10, w.dtl);
    MMVAE_TRY(mmvae_check_launch("logsoftmax_bwd32"));
    return txt_dec_bwd32(P, w.z, w.dtl, 1, dz, s);
}
