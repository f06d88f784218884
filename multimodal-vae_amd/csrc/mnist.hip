// MNIST MMVAE (mnist/model.py:14-185, mnist/train.py:64-81,131-147): Linear -> BatchNorm1d -> ReLU stacks, label
// embedding, product of experts, BCE + NLL + KL/(B*784/3), all on the shared GEMM / BN / latent kernels.
//
// Pass structure (train.py:136-147): (image,label), (image), (label), all lambdas 1.  The image encoder sees the same
// input in passes 1 and 2 and the label encoder in passes 1 and 3 (no dropout anywhere in this model), so each runs
// ONCE with its BatchNorm running statistics updated twice and receives the SUM of the two passes' output gradients
// (exact: BatchNorm backward is linear in the upstream gradient for fixed statistics).  Decoders run on 3*B rows with
// statistics per pass group.
#include "mnist.h"
#include "plan_base.h"
#include <cstring>
#include <algorithm>

#include "mnist_plan.h"

namespace {

void build(MnistPlan& P) {
    const int D = P.D;
    P.ldz = round_up(D + 1, 8);
    auto lin = [&](const std::string& n, int o, int i) { add_param(P, n + ".weight", {o, i}); add_param(P, n + ".bias", {o}); };
    auto bnp = [&](const std::string& n, int c) { add_param(P, n + ".weight", {c}); add_param(P, n + ".bias", {c}); };
    lin("image_encoder.net.0", 400, 784); bnp("image_encoder.net.1", 400);
    lin("image_encoder.net.3", 200, 400); bnp("image_encoder.net.4", 200);
    lin("image_encoder.net.6", 2 * D, 200);
    lin("image_decoder.net.0", 200, D); bnp("image_decoder.net.1", 200);
    lin("image_decoder.net.3", 400, 200); bnp("image_decoder.net.4", 400);
    lin("image_decoder.net.6", 784, 400);
    add_param(P, "text_encoder.net.0.weight", {10, 50}); bnp("text_encoder.net.1", 50);
    lin("text_encoder.net.3", 2 * D, 50);
    lin("text_decoder.net.0", 10, D); bnp("text_decoder.net.1", 10);
    lin("text_decoder.net.3", 10, 10);
    const char* bnn[6] = {"image_encoder.net.1", "image_encoder.net.4", "image_decoder.net.1", "image_decoder.net.4",
                          "text_encoder.net.1", "text_decoder.net.1"};
    const int bnc[6] = {400, 200, 200, 400, 50, 10};
    long long so = 0;
    for (int i = 0; i < 6; ++i) {
        P.bn[i] = BnL{off(P, std::string(bnn[i]) + ".weight"), off(P, std::string(bnn[i]) + ".bias"), bnc[i], so, i};
        P.bn_names.push_back(bnn[i]); P.bn_list.push_back(P.bn[i]);
        so += 2 * bnc[i];
    }
    P.emb_off = off(P, "text_encoder.net.0.weight");
    mlp_lin_init(P, P.ie[0], "image_encoder.net.0", 400, 784, 0, false);
    mlp_lin_init(P, P.ie[1], "image_encoder.net.3", 200, 400, 1, true);
    mlp_lin_init(P, P.ie[2], "image_encoder.net.6", 2 * D, 200, -1, true);
    mlp_lin_init(P, P.id[0], "image_decoder.net.0", 200, D, 2, true, P.ldz);    // operand is z_bf
    mlp_lin_init(P, P.id[1], "image_decoder.net.3", 400, 200, 3, true);
    mlp_lin_init(P, P.id[2], "image_decoder.net.6", 784, 400, -1, true);
    mlp_lin_init(P, P.te_lin, "text_encoder.net.3", 2 * D, 50, -1, true);
    mlp_lin_init(P, P.td[0], "text_decoder.net.0", 10, D, 5, true, P.ldz);
    mlp_lin_init(P, P.td[1], "text_decoder.net.3", 10, 10, -1, true);
}

void carve(MnistPlan& P, Workspace& ws) {
    MnistPlan::W& w = P.w;
    const size_t B = P.B, D = P.D, B3 = 3 * B;
    const int SS = MMVAE_STAT_SLOTS;
    const int bnc[6] = {400, 200, 200, 400, 50, 10};
    const int bng[6] = {1, 1, 3, 3, 1, 3};
    char* z0 = ws.take<char>(0);
    for (int i = 0; i < 6; ++i) { w.st[i] = ws.take<float2>(bng[i] * SS * bnc[i]); w.red[i] = ws.take<float2>(bng[i] * SS * bnc[i]); }
    w.sums = ws.take<float>(16 * MMVAE_LOSS_SLOTS);
    w.dz_img = ws.take<float>(B3 * D); w.dz_txt = ws.take<float>(B3 * D);
    char* z1 = ws.take<char>(0);
    w.zero_begin = z0; w.zero_bytes = (size_t)(z1 - z0);
    for (int i = 0; i < 6; ++i) { w.aff[i] = ws.take<float2>(bng[i] * bnc[i]); w.mr[i] = ws.take<float2>(bng[i] * bnc[i]); }
    w.x_bf = ws.take<bf16>(B * 784);
    w.r_ie[0] = ws.take<bf16>(B * 400); w.a_ie[0] = ws.take<bf16>(B * 400);
    w.r_ie[1] = ws.take<bf16>(B * 200); w.a_ie[1] = ws.take<bf16>(B * 200);
    w.encout = ws.take<float>(B * 2 * D);
    w.r_te = ws.take<bf16>(B * 56); w.a_te = ws.take<bf16>(B * 56); w.txtout = ws.take<float>(B * 2 * D);
    w.eps = ws.take<float>(B3 * D); w.mu = ws.take<float>(B3 * D); w.logvar = ws.take<float>(B3 * D);
    w.z_f32 = ws.take<float>(B3 * D); w.z_bf = ws.take<bf16>(B3 * P.ldz);
    w.r_id[0] = ws.take<bf16>(B3 * 200); w.a_id[0] = ws.take<bf16>(B3 * 200);
    w.r_id[1] = ws.take<bf16>(B3 * 400); w.a_id[1] = ws.take<bf16>(B3 * 400);
    w.logits = ws.take<float>(B3 * 784); w.dlogit = ws.take<float>(B3 * 784); w.dlogit_bf = ws.take<bf16>(B3 * 784);
    w.r_td = ws.take<bf16>(B3 * 16); w.a_td = ws.take<bf16>(B3 * 16);
    w.tlogits = ws.take<float>(B3 * 10); w.words = ws.take<float>(B3 * 10); w.dtl = ws.take<bf16>(B3 * 16);
    w.d_id[0] = ws.take<bf16>(B3 * 200); w.d_id[1] = ws.take<bf16>(B3 * 400); w.d_td = ws.take<bf16>(B3 * 16);
    w.d_encout = ws.take<bf16>(B * 2 * D); w.d_txtout_bf = ws.take<bf16>(B * 2 * D);
    w.d_ie[0] = ws.take<bf16>(B * 400); w.d_ie[1] = ws.take<bf16>(B * 200); w.d_te = ws.take<bf16>(B * 56);
    P.sk_cnt = ws.take<unsigned>(1024);
    P.sk_floats = (size_t)256 * 128 * 128;
    P.sk_buf = ws.take<float>(P.sk_floats);
}

BnTabs tabs(MnistPlan& P, int bi) { MnistPlan::W& w = P.w; return BnTabs{w.st[bi], w.red[bi], w.aff[bi], w.mr[bi]}; }
int lin_fwd(MnistPlan& P, const MlpLin& L, const bf16* A, int rows, int groups, bf16* out_bf, float* out_f, float2* stats, hipStream_t s) {
    return mlp_fwd(P, L, A, rows, groups, out_bf, out_f, stats, s);
}
int lin_dgrad(MnistPlan& P, const MlpLin& L, const bf16* dY, int rows, int groups, bf16* out_bf, float* out_f, int out_ld,
              const bf16* r_prev, int pbn, hipStream_t s) {
    BnTabs t{};
    if (r_prev) t = tabs(P, pbn);
    return mlp_dgrad(P, L, dY, rows, groups, out_bf, out_f, out_ld, r_prev, r_prev ? &t : nullptr, ACT_RELU, s);
}
int lin_wgrad(MnistPlan& P, const MlpLin& L, const bf16* dY, const bf16* A, int rows, hipStream_t s) { return mlp_wgrad(P, L, dY, A, rows, s); }
int mn_bn_act(MnistPlan& P, int bi, const bf16* r, bf16* a, int rows, int groups, int ld, int updates, int training, hipStream_t s) {
    return bn1d_act(P, P.bn[bi], tabs(P, bi), r, a, rows, groups, ld, updates, training, ACT_RELU, s);
}
int mn_bn_bwd(MnistPlan& P, int bi, bf16* d, const bf16* r, int rows, int groups, int ld, hipStream_t s) {
    return bn1d_bwd(P, P.bn[bi], tabs(P, bi), d, r, rows, groups, ld, s);
}

}  // namespace

MnistPlan* mnist_create(int D, int B, int precision) {
    if (D < 4 || D > 124 || D % 4 != 0 || B < 1) { mmvae_set_error("mnist_create: need n_latents in 4..124, a multiple of 4, and batch >= 1"); return nullptr; }
    MnistPlan* P = new MnistPlan();
    P->D = D; P->B = B;
    // precision: 0 = fp32 (the reference's own arithmetic; default), 1 = bf16 MFMA operands (the engine's conv path)
    if (precision < 0) { const char* e = getenv("MMVAE_MNIST_PRECISION"); precision = (e && !strcmp(e, "bf16")) ? 1 : 0; }
    P->f32 = precision == 0;
    P->no_pack = P->f32;
    build(*P);
    Workspace ws(nullptr, 0);
    carve(*P, ws);
    P->ws_bytes = std::max(ws.used(), mnist_f32_workspace_bytes(*P));
    return P;
}
void mnist_destroy(MnistPlan* P) { delete P; }
int mnist_is_f32(const MnistPlan* P) { return P->f32 ? 1 : 0; }
PlanBase* mnist_base(MnistPlan* P) { return P; }

int mnist_step(MnistPlan* Pp, const MnistStepIO& io, int training, int do_backward, hipStream_t s) {
    MMVAE_TRY(check_bound(Pp));
    MnistPlan& P = *Pp;
    if (P.f32) return mnist_f32_step(P, io, training, do_backward, s);
    MMVAE_REQUIRE(io.ws && io.ws_bytes >= P.ws_bytes, "mnist step: workspace too small");
    MMVAE_REQUIRE(io.image && io.label && io.sums, "mnist step: image/label/sums must be given");
    Workspace wsp(io.ws, io.ws_bytes);
    carve(P, wsp);
    MnistPlan::W& w = P.w;
    const int B = P.B, D = P.D, B3 = 3 * B, D2 = 2 * D;
    const float* eps = io.eps;
    StepBeginArgs sb{};
    sb.zero_ptr[0] = w.zero_begin; sb.zero_bytes[0] = w.zero_bytes;
    if (do_backward) {
        sb.zero_ptr[1] = P.buf.gpk; sb.zero_bytes[1] = (size_t)P.gk.mat_elems * sizeof(float);
        sb.zero_ptr[2] = P.buf.grads; sb.zero_bytes[2] = (size_t)(P.nparams / 4) * 16;
    }
    sb.seed = io.seed; sb.step = io.step_ctr;
    if (training && !eps) { sb.eps = w.eps; sb.n_eps = (long long)B3 * D; eps = w.eps; }
    MMVAE_TRY(launch_step_begin(sb, s));
    if (do_backward && P.nparams % 4 != 0)
        MMVAE_TRY(launch_fill_zero(P.buf.grads + (P.nparams / 4) * 4, (size_t)(P.nparams % 4) * sizeof(float), s));
    MMVAE_TRY(ensure_streams(P));
    P.wgrad_forked = false;
    const int sk[3] = {io.pass_skip[0] != 0, io.pass_skip[1] != 0, io.pass_skip[2] != 0};
    P.dec_skip_mask = (unsigned)(sk[0] | (sk[1] << 1) | (sk[2] << 2));
    const int img_updates = 2 - sk[0] - sk[1], txt_updates = 2 - sk[0] - sk[2];
    // ---- image encoder (mnist/model.py:99-118), once for passes 1 and 2
    MMVAE_TRY(launch_cast_bf16(io.image, (long long)B * 784, w.x_bf, s));
    MMVAE_TRY(lin_fwd(P, P.ie[0], w.x_bf, B, 1, w.r_ie[0], nullptr, training ? w.st[0] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 0, w.r_ie[0], w.a_ie[0], B, 1, 400, img_updates, training, s));
    MMVAE_TRY(lin_fwd(P, P.ie[1], w.a_ie[0], B, 1, w.r_ie[1], nullptr, training ? w.st[1] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 1, w.r_ie[1], w.a_ie[1], B, 1, 200, img_updates, training, s));
    MMVAE_TRY(lin_fwd(P, P.ie[2], w.a_ie[1], B, 1, nullptr, w.encout, nullptr, s));
    // ---- label encoder (mnist/model.py:136-153), once for passes 1 and 3
    MMVAE_TRY(launch_embed_gather_stats(P.buf.params + P.emb_off, 50, io.label, B, B, B, w.r_te, 56, training ? w.st[4] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 4, w.r_te, w.a_te, B, 1, 56, txt_updates, training, s));
    MMVAE_TRY(lin_fwd(P, P.te_lin, w.a_te, B, 1, nullptr, w.txtout, nullptr, s));
    // ---- product of experts, reparametrisation, KL
    Latent3Args la{};
    la.B = B; la.D = D; la.img_out = w.encout; la.img_out_b = w.encout; la.txt_out = w.txtout; la.eps = eps;
    la.mu = io.mu ? io.mu : w.mu; la.logvar = io.logvar ? io.logvar : w.logvar;
    la.z_f32 = w.z_f32; la.z_bf = w.z_bf; la.ldz = P.ldz; la.kl_sum = w.sums + 8; la.training = training;
    MMVAE_TRY(launch_latent3_fwd(la, s));
    // ---- image decoder on 3B rows, BatchNorm per pass (mnist/model.py:121-133)
    MMVAE_TRY(lin_fwd(P, P.id[0], w.z_bf, B3, 3, w.r_id[0], nullptr, training ? w.st[2] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 2, w.r_id[0], w.a_id[0], B3, 3, 200, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.id[1], w.a_id[0], B3, 3, w.r_id[1], nullptr, training ? w.st[3] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 3, w.r_id[1], w.a_id[1], B3, 3, 400, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.id[2], w.a_id[1], B3, 1, nullptr, w.logits, nullptr, s));
    BceArgs bc{};
    bc.logits = w.logits; bc.ldl = 1; bc.target = io.image; bc.G = 3; bc.B = B; bc.C = 1; bc.H = 28; bc.W = 28;
    bc.recon = io.recon_image; bc.dlogit = do_backward ? w.dlogit : nullptr; bc.loss_sum = w.sums;
    for (int k = 0; k < 3; ++k) bc.coef[k] = sk[k] ? 0.f : io.lambda_xy[k] / (float)(B * 784);
    MMVAE_TRY(launch_sigmoid_bce(bc, s));
    // ---- label decoder (mnist/model.py:156-170) + NLL
    MMVAE_TRY(lin_fwd(P, P.td[0], w.z_bf, B3, 3, w.r_td, nullptr, training ? w.st[5] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 5, w.r_td, w.a_td, B3, 3, 16, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.td[1], w.a_td, B3, 1, nullptr, w.tlogits, nullptr, s));
    LogSoftmaxNllArgs ls{};
    ls.logits = w.tlogits; ls.rows = B3; ls.classes = 10; ls.words = io.recon_text ? io.recon_text : w.words;
    ls.target = io.label; ls.target_rows = B; ls.rows_per_group = B; ls.nll_sum = w.sums + 4;
    ls.dlogits = do_backward ? w.dtl : nullptr; ls.ld_d = 16;
    for (int k = 0; k < 3; ++k) ls.coef[k] = sk[k] ? 0.f : io.lambda_yx[k] / (float)B;
    MMVAE_TRY(launch_logsoftmax_nll(ls, s));
    hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums);
    MMVAE_TRY(mmvae_check_launch("sum_slots"));
    if (!do_backward) return MMVAE_OK;

    // =============================== backward ===============================
    // every kernel of this model is launch-latency sized (~50 launches of a few us): forking the weight gradients to
    // side streams costs more in event edges than it overlaps (measured 0.405 ms forked vs 0.387 ms in-order at B=128)
    P.wgrad_forked = false;
    int rc = MMVAE_OK;
    auto TRY = [&](int r) { if (rc == MMVAE_OK) rc = r; };
    float* G = P.buf.grads;
    // label decoder
    TRY(launch_colsum_bf16(w.dtl, 16, B3, 10, G + P.td[1].b_off, s));
    TRY(lin_wgrad(P, P.td[1], w.dtl, w.a_td, B3, s));
    TRY(lin_dgrad(P, P.td[1], w.dtl, B3, 3, w.d_td, nullptr, 16, w.r_td, 5, s));
    TRY(mn_bn_bwd(P, 5, w.d_td, w.r_td, B3, 3, 16, s));
    TRY(lin_wgrad(P, P.td[0], w.d_td, w.z_bf, B3, s));
    TRY(lin_dgrad(P, P.td[0], w.d_td, B3, 1, nullptr, w.dz_txt, D, nullptr, -1, s));
    // image decoder
    TRY(launch_cast_bf16(w.dlogit, (long long)B3 * 784, w.dlogit_bf, s));
    TRY(launch_colsum_f32(w.dlogit, B3, 784, G + P.id[2].b_off, s));
    TRY(lin_wgrad(P, P.id[2], w.dlogit_bf, w.a_id[1], B3, s));
    TRY(lin_dgrad(P, P.id[2], w.dlogit_bf, B3, 3, w.d_id[1], nullptr, 400, w.r_id[1], 3, s));
    TRY(mn_bn_bwd(P, 3, w.d_id[1], w.r_id[1], B3, 3, 400, s));
    TRY(lin_wgrad(P, P.id[1], w.d_id[1], w.a_id[0], B3, s));
    TRY(lin_dgrad(P, P.id[1], w.d_id[1], B3, 3, w.d_id[0], nullptr, 200, w.r_id[0], 2, s));
    TRY(mn_bn_bwd(P, 2, w.d_id[0], w.r_id[0], B3, 3, 200, s));
    TRY(lin_wgrad(P, P.id[0], w.d_id[0], w.z_bf, B3, s));
    TRY(lin_dgrad(P, P.id[0], w.d_id[0], B3, 1, nullptr, w.dz_img, D, nullptr, -1, s));
    // latent block
    Latent3BwdArgs lb{};
    lb.f = la; lb.dz_a = w.dz_img; lb.dz_b = w.dz_txt;
    for (int k = 0; k < 3; ++k) lb.kl_coef[k] = sk[k] ? 0.f : io.kl_coef;
    lb.d_img_out_bf = w.d_encout; lb.sum_img_variants = 1; lb.d_img_bias = G + P.ie[2].b_off;
    lb.d_txt_out = nullptr; lb.d_txt_out_bf = w.d_txtout_bf; lb.d_txt_bias = G + P.te_lin.b_off;
    TRY(launch_latent3_bwd(lb, s));
    // image encoder
    TRY(lin_wgrad(P, P.ie[2], w.d_encout, w.a_ie[1], B, s));
    TRY(lin_dgrad(P, P.ie[2], w.d_encout, B, 1, w.d_ie[1], nullptr, 200, w.r_ie[1], 1, s));
    TRY(mn_bn_bwd(P, 1, w.d_ie[1], w.r_ie[1], B, 1, 200, s));
    TRY(lin_wgrad(P, P.ie[1], w.d_ie[1], w.a_ie[0], B, s));
    TRY(lin_dgrad(P, P.ie[1], w.d_ie[1], B, 1, w.d_ie[0], nullptr, 400, w.r_ie[0], 0, s));
    TRY(mn_bn_bwd(P, 0, w.d_ie[0], w.r_ie[0], B, 1, 400, s));
    TRY(lin_wgrad(P, P.ie[0], w.d_ie[0], w.x_bf, B, s));
    // label encoder
    TRY(lin_wgrad(P, P.te_lin, w.d_txtout_bf, w.a_te, B, s));
    TRY(lin_dgrad(P, P.te_lin, w.d_txtout_bf, B, 1, w.d_te, nullptr, 56, w.r_te, 4, s));
    TRY(mn_bn_bwd(P, 4, w.d_te, w.r_te, B, 1, 56, s));
    TRY(launch_embed_scatter_add(w.d_te, 56, 50, io.label, B, G + P.emb_off, s));
    P.wgrad_forked = false;
    MMVAE_TRY(rc);
    MMVAE_TRY(edge(P, P.st_wgrad, s));
    if (P.st_wgrad2 != P.st_wgrad) MMVAE_TRY(edge(P, P.st_wgrad2, s));
    return launch_unpack_grads(P.buf.gdesc_dev, P.gk.d.data(), (int)P.gk.d.size(), P.buf.gpk, P.buf.gpk_vec, P.buf.grads, s);
}

// ================================================================== granular modules (drop-in nn.Module forwards)
// Each call works on B rows (one BatchNorm group) in the workspace it is handed; the backward must get the workspace
// of its forward.  Parameter gradients accumulate into the bound `grads` (the caller zeroes them).
namespace {
int mn_use_ws(MnistPlan* P, void* ws, size_t bytes) {
    MMVAE_TRY(check_bound(P));
    MMVAE_REQUIRE(ws != nullptr && bytes >= P->ws_bytes, "workspace too small (%zu < %zu)", bytes, P->ws_bytes);
    Workspace w(ws, bytes);
    carve(*P, w);
    P->wgrad_forked = false;
    P->dec_skip_mask = 0;
    return MMVAE_OK;
}
int mn_zero(MnistPlan& P, bool backward, hipStream_t s) {
    MMVAE_TRY(launch_fill_zero(P.w.zero_begin, P.w.zero_bytes, s));
    if (backward) MMVAE_TRY(launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s));
    return MMVAE_OK;
}
int mn_unpack(MnistPlan& P, hipStream_t s) {
    return launch_unpack_grads(P.buf.gdesc_dev, P.gk.d.data(), (int)P.gk.d.size(), P.buf.gpk, P.buf.gpk_vec, P.buf.grads, s);
}
// dlogits[r][c] = d_logp[r][c] - softmax[r][c] * sum_c d_logp[r][c]   (log_softmax backward), bf16 rows of stride ld
__global__ void logsoftmax_bwd_kernel(const float* d_logp, const float* logp, int rows, int classes, bf16* out, int ld) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float sum = 0.f;
    for (int c = 0; c < classes; ++c) sum += d_logp[(size_t)r * classes + c];
    for (int c = 0; c < ld; ++c)
        out[(size_t)r * ld + c] = (bf16)(c < classes ? d_logp[(size_t)r * classes + c] - expf(logp[(size_t)r * classes + c]) * sum : 0.f);
}
}  // namespace

int mnist_image_encoder_fwd(MnistPlan* Pp, void* ws, size_t wsb, const float* image, int training, float* out, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_image_encoder_fwd(*Pp, ws, wsb, image, training, out, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B;
    MMVAE_TRY(mn_zero(P, false, s));
    MMVAE_TRY(launch_cast_bf16(image, (long long)B * 784, w.x_bf, s));
    MMVAE_TRY(lin_fwd(P, P.ie[0], w.x_bf, B, 1, w.r_ie[0], nullptr, training ? w.st[0] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 0, w.r_ie[0], w.a_ie[0], B, 1, 400, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.ie[1], w.a_ie[0], B, 1, w.r_ie[1], nullptr, training ? w.st[1] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 1, w.r_ie[1], w.a_ie[1], B, 1, 200, 1, training, s));
    return lin_fwd(P, P.ie[2], w.a_ie[1], B, 1, nullptr, out, nullptr, s);
}
int mnist_image_encoder_bwd(MnistPlan* Pp, void* ws, size_t wsb, const float* d_out, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_image_encoder_bwd(*Pp, ws, wsb, d_out, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B, D2 = 2 * P.D;
    MMVAE_TRY(launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s));
    MMVAE_TRY(launch_cast_bf16(d_out, (long long)B * D2, w.d_encout, s));
    MMVAE_TRY(launch_colsum_f32(d_out, B, D2, P.buf.grads + P.ie[2].b_off, s));
    MMVAE_TRY(lin_wgrad(P, P.ie[2], w.d_encout, w.a_ie[1], B, s));
    MMVAE_TRY(lin_dgrad(P, P.ie[2], w.d_encout, B, 1, w.d_ie[1], nullptr, 200, w.r_ie[1], 1, s));
    MMVAE_TRY(mn_bn_bwd(P, 1, w.d_ie[1], w.r_ie[1], B, 1, 200, s));
    MMVAE_TRY(lin_wgrad(P, P.ie[1], w.d_ie[1], w.a_ie[0], B, s));
    MMVAE_TRY(lin_dgrad(P, P.ie[1], w.d_ie[1], B, 1, w.d_ie[0], nullptr, 400, w.r_ie[0], 0, s));
    MMVAE_TRY(mn_bn_bwd(P, 0, w.d_ie[0], w.r_ie[0], B, 1, 400, s));
    MMVAE_TRY(lin_wgrad(P, P.ie[0], w.d_ie[0], w.x_bf, B, s));
    return mn_unpack(P, s);
}
int mnist_image_decoder_fwd(MnistPlan* Pp, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_image_decoder_fwd(*Pp, ws, wsb, z, training, recon, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B;
    MMVAE_TRY(mn_zero(P, false, s));
    hipLaunchKernelGGL(cast_z_kernel, dim3(ceil_div(B * P.ldz, 256)), dim3(256), 0, s, z, B, P.D, w.z_bf, P.ldz);
    MMVAE_TRY(mmvae_check_launch("cast_z"));
    MMVAE_TRY(lin_fwd(P, P.id[0], w.z_bf, B, 1, w.r_id[0], nullptr, training ? w.st[2] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 2, w.r_id[0], w.a_id[0], B, 1, 200, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.id[1], w.a_id[0], B, 1, w.r_id[1], nullptr, training ? w.st[3] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 3, w.r_id[1], w.a_id[1], B, 1, 400, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.id[2], w.a_id[1], B, 1, nullptr, w.logits, nullptr, s));
    const long long n = (long long)B * 784;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, w.logits, n, recon);
    return mmvae_check_launch("sigmoid");
}
int mnist_image_decoder_bwd(MnistPlan* Pp, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_image_decoder_bwd(*Pp, ws, wsb, d_recon, recon, dz, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B;
    MMVAE_TRY(launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s));
    const long long n = (long long)B * 784;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, d_recon, recon, n, w.dlogit);
    MMVAE_TRY(mmvae_check_launch("sigmoid_bwd"));
    MMVAE_TRY(launch_cast_bf16(w.dlogit, n, w.dlogit_bf, s));
    MMVAE_TRY(launch_colsum_f32(w.dlogit, B, 784, P.buf.grads + P.id[2].b_off, s));
    MMVAE_TRY(lin_wgrad(P, P.id[2], w.dlogit_bf, w.a_id[1], B, s));
    MMVAE_TRY(lin_dgrad(P, P.id[2], w.dlogit_bf, B, 1, w.d_id[1], nullptr, 400, w.r_id[1], 3, s));
    MMVAE_TRY(mn_bn_bwd(P, 3, w.d_id[1], w.r_id[1], B, 1, 400, s));
    MMVAE_TRY(lin_wgrad(P, P.id[1], w.d_id[1], w.a_id[0], B, s));
    MMVAE_TRY(lin_dgrad(P, P.id[1], w.d_id[1], B, 1, w.d_id[0], nullptr, 200, w.r_id[0], 2, s));
    MMVAE_TRY(mn_bn_bwd(P, 2, w.d_id[0], w.r_id[0], B, 1, 200, s));
    MMVAE_TRY(lin_wgrad(P, P.id[0], w.d_id[0], w.z_bf, B, s));
    MMVAE_TRY(lin_dgrad(P, P.id[0], w.d_id[0], B, 1, nullptr, dz, P.D, nullptr, -1, s));
    return mn_unpack(P, s);
}
int mnist_text_encoder_fwd(MnistPlan* Pp, void* ws, size_t wsb, const long long* label, int training, float* out, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_text_encoder_fwd(*Pp, ws, wsb, label, training, out, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B;
    MMVAE_TRY(mn_zero(P, false, s));
    MMVAE_TRY(launch_embed_gather_stats(P.buf.params + P.emb_off, 50, label, B, B, B, w.r_te, 56, training ? w.st[4] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 4, w.r_te, w.a_te, B, 1, 56, 1, training, s));
    return lin_fwd(P, P.te_lin, w.a_te, B, 1, nullptr, out, nullptr, s);
}
int mnist_text_encoder_bwd(MnistPlan* Pp, void* ws, size_t wsb, const long long* label, const float* d_out, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_text_encoder_bwd(*Pp, ws, wsb, label, d_out, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B, D2 = 2 * P.D;
    MMVAE_TRY(launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s));
    MMVAE_TRY(launch_cast_bf16(d_out, (long long)B * D2, w.d_txtout_bf, s));
    MMVAE_TRY(launch_colsum_f32(d_out, B, D2, P.buf.grads + P.te_lin.b_off, s));
    MMVAE_TRY(lin_wgrad(P, P.te_lin, w.d_txtout_bf, w.a_te, B, s));
    MMVAE_TRY(lin_dgrad(P, P.te_lin, w.d_txtout_bf, B, 1, w.d_te, nullptr, 56, w.r_te, 4, s));
    MMVAE_TRY(mn_bn_bwd(P, 4, w.d_te, w.r_te, B, 1, 56, s));
    MMVAE_TRY(launch_embed_scatter_add(w.d_te, 56, 50, label, B, P.buf.grads + P.emb_off, s));
    return mn_unpack(P, s);
}
int mnist_text_decoder_fwd(MnistPlan* Pp, void* ws, size_t wsb, const float* z, int training, float* logp, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_text_decoder_fwd(*Pp, ws, wsb, z, training, logp, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B;
    MMVAE_TRY(mn_zero(P, false, s));
    hipLaunchKernelGGL(cast_z_kernel, dim3(ceil_div(B * P.ldz, 256)), dim3(256), 0, s, z, B, P.D, w.z_bf, P.ldz);
    MMVAE_TRY(mmvae_check_launch("cast_z"));
    MMVAE_TRY(lin_fwd(P, P.td[0], w.z_bf, B, 1, w.r_td, nullptr, training ? w.st[5] : nullptr, s));
    MMVAE_TRY(mn_bn_act(P, 5, w.r_td, w.a_td, B, 1, 16, 1, training, s));
    MMVAE_TRY(lin_fwd(P, P.td[1], w.a_td, B, 1, nullptr, w.tlogits, nullptr, s));
    LogSoftmaxNllArgs ls{};
    ls.logits = w.tlogits; ls.rows = B; ls.classes = 10; ls.words = logp; ls.rows_per_group = B; ls.target_rows = B;
    return launch_logsoftmax_nll(ls, s);
}
int mnist_text_decoder_bwd(MnistPlan* Pp, void* ws, size_t wsb, const float* d_logp, const float* logp, float* dz, hipStream_t s) {
    if (Pp && Pp->f32) { MMVAE_TRY(check_bound(Pp)); return mnist_f32_text_decoder_bwd(*Pp, ws, wsb, d_logp, logp, dz, s); }
    MMVAE_TRY(mn_use_ws(Pp, ws, wsb));
    MnistPlan& P = *Pp; MnistPlan::W& w = P.w; const int B = P.B;
    MMVAE_TRY(launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s));
    hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, d_logp, logp, B, 10, w.dtl, 16);
    MMVAE_TRY(mmvae_check_launch("logsoftmax_bwd"));
    MMVAE_TRY(launch_colsum_bf16(w.dtl, 16, B, 10, P.buf.grads + P.td[1].b_off, s));
    MMVAE_TRY(lin_wgrad(P, P.td[1], w.dtl, w.a_td, B, s));
    MMVAE_TRY(lin_dgrad(P, P.td[1], w.dtl, B, 1, w.d_td, nullptr, 16, w.r_td, 5, s));
    MMVAE_TRY(mn_bn_bwd(P, 5, w.d_td, w.r_td, B, 1, 16, s));
    MMVAE_TRY(lin_wgrad(P, P.td[0], w.d_td, w.z_bf, B, s));
    MMVAE_TRY(lin_dgrad(P, P.td[0], w.d_td, B, 1, nullptr, dz, P.D, nullptr, -1, s));
    return mn_unpack(P, s);
}
