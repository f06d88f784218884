// MultiMNIST MMVAE: layer plans, weight-pack tables and the kernel sequences of the ELBO step.
// Reference: multimnist/model.py:21-93 (MultimodalVAE), :150-216 (image enc/dec), :219-307 (text enc/dec),
//            multimnist/train.py:69-87 (loss_function), :146-173 (3-pass step).
#include "multimnist.h"
#include <memory>
#include "mlp_tail.h"
#include "plan_base.h"
#include "thin.h"
#include <cstring>

namespace {
constexpr int IMG = 50, NPIX = 2500;
}

struct MMPlan : PlanBase {
    int ldz, kx, kz;
    ConvL conv[4], convT[4];
    LinL fc[3], up;
    BnL bn[6];
    // text packs
    struct GruIdx { int wih, whh, wihT, whhT, g_wih, g_whh; long long bih, bhh; } te_f, te_r, td0, td1;
    const float* step_image = nullptr;      // the image batch of the running fused step (conv1's weight gradient rebuilds its patches)
    hipEvent_t ev_dz = nullptr;             // fused step: completion of the text decoder's backward kernel (what the main chain joins on)
    hipEvent_t ev_txtgrads = nullptr;       // fused step: every gradient of text_decoder.* is complete (recorded on the second-modality stream)
    int te_h2p, te_h2pT, g_te_h2p, td_z2h, td_z2hT, g_td_z2h, td_h2o, td_h2oT, g_td_h2o;
    // fused classifier tail (mlp_tail.hip; n_latents = 100 only): fragment-major copies of classifier.3 / classifier.6
    bool mlp_tail = false;
    int mt_w2, mt_w3, mt_w3t, mt_w2t;
    // ---- workspace pointers
    struct W {
        char* zero_begin; size_t zero_bytes;
        float2 *st_e[3], *red_e[3], *st_d[3], *red_d[3];
        float* sums; float* dz_img; float* dz_txt;
        float2 *aff_e[3], *mr_e[3], *aff_d[3], *mr_d[3];
        bf16 *patches1, *r1, *r2, *r3, *r4, *y1, *y2;
        bf16 *a1, *a2, *a3, *a4, *ay1, *ay2, *au, *aq1, *aq2, *aq3;
        float* encout; uint8_t *m1, *m2, *gkeep;
        float *txtout, *te_gates_f, *te_gates_r; bf16 *te_x, *te_hprev, *te_hsum;
        float *eps, *mu, *logvar, *z_f32; bf16* z_bf;
        bf16 *u, *q1, *q2, *q3; float *logits, *recon, *dlogit;
        float *words, *dwords, *td_gates; long long* tokens;
        bf16 *td_x0, *td_h0p, *td_mid, *td_h1p, *td_hz, *td_zbf;
        bf16 *dgi0, *dgh0, *dgi1, *dgh1, *dlogit_bf, *dhinit;
        bf16 *patches4, *d3, *d2, *d1, *du;
        bf16 *d_encout; float* d_txtout; bf16* te_dout_bf; bf16 *te_dgi_f, *te_dgh_f, *te_dgi_r;
        bf16 *dy2, *dy1, *db4, *dr4, *d3e, *d2e, *d1e;
        bf16 *d3r, *d2r, *d1r, *d3er, *d2er;       // BatchNorm-backward outputs kept apart from db (fused staging: see dec_bwd / enc_bwd)
        float* tmp_f32;
        float* slab; size_t slab_floats;
    } w;
};

namespace {

void build_params(MMPlan& P) {
    const int D = P.D;
    auto bn = [&](const std::string& n, int c) { add_param(P, n + ".weight", {c}); add_param(P, n + ".bias", {c}); };
    auto gru = [&](const std::string& n, int inp, const char* sfx) {
        add_param(P, n + ".weight_ih_" + sfx, {300, inp}); add_param(P, n + ".weight_hh_" + sfx, {300, 100});
        add_param(P, n + ".bias_ih_" + sfx, {300}); add_param(P, n + ".bias_hh_" + sfx, {300});
    };
    add_param(P, "image_encoder.features.0.weight", {32, 1, 4, 4});
    add_param(P, "image_encoder.features.2.weight", {64, 32, 4, 4}); bn("image_encoder.features.3", 64);
    add_param(P, "image_encoder.features.5.weight", {128, 64, 4, 4}); bn("image_encoder.features.6", 128);
    add_param(P, "image_encoder.features.8.weight", {256, 128, 4, 4}); bn("image_encoder.features.9", 256);
    add_param(P, "image_encoder.classifier.0.weight", {400, 1024}); add_param(P, "image_encoder.classifier.0.bias", {400});
    add_param(P, "image_encoder.classifier.3.weight", {200, 400}); add_param(P, "image_encoder.classifier.3.bias", {200});
    add_param(P, "image_encoder.classifier.6.weight", {2 * D, 200}); add_param(P, "image_encoder.classifier.6.bias", {2 * D});
    add_param(P, "image_decoder.upsample.0.weight", {1024, D}); add_param(P, "image_decoder.upsample.0.bias", {1024});
    add_param(P, "image_decoder.hallucinate.0.weight", {256, 128, 4, 4}); bn("image_decoder.hallucinate.1", 128);
    add_param(P, "image_decoder.hallucinate.3.weight", {128, 64, 4, 4}); bn("image_decoder.hallucinate.4", 64);
    add_param(P, "image_decoder.hallucinate.6.weight", {64, 32, 5, 5}); bn("image_decoder.hallucinate.7", 32);
    add_param(P, "image_decoder.hallucinate.9.weight", {32, 1, 4, 4});
    add_param(P, "text_encoder.embed.weight", {12, 100});
    gru("text_encoder.gru", 100, "l0"); gru("text_encoder.gru", 100, "l0_reverse");
    add_param(P, "text_encoder.h2p.weight", {2 * D, 100}); add_param(P, "text_encoder.h2p.bias", {2 * D});
    add_param(P, "text_decoder.embed.weight", {12, 100});
    add_param(P, "text_decoder.z2h.weight", {100, D}); add_param(P, "text_decoder.z2h.bias", {100});
    gru("text_decoder.gru", 100 + D, "l0"); gru("text_decoder.gru", 100, "l1");
    add_param(P, "text_decoder.h2o.weight", {12, 100 + D}); add_param(P, "text_decoder.h2o.bias", {12});
}

void build_plan(MMPlan& P) {
    const int D = P.D;
    P.ldz = round_up(D + 1, 8); P.kz = round_up(D, 32); P.kx = round_up(100 + D, 32);
    build_params(P);
    // BatchNorm tables (state_dict order)
    const char* bnn[6] = {"image_encoder.features.3", "image_encoder.features.6", "image_encoder.features.9",
                          "image_decoder.hallucinate.1", "image_decoder.hallucinate.4", "image_decoder.hallucinate.7"};
    const int bnc[6] = {64, 128, 256, 128, 64, 32};
    long long so = 0;
    for (int i = 0; i < 6; ++i) {
        P.bn[i] = BnL{off(P, std::string(bnn[i]) + ".weight"), off(P, std::string(bnn[i]) + ".bias"), bnc[i], so, i};
        so += 2 * bnc[i];
    }
    // encoder convs (multimnist/model.py:160-169)
    build_conv(P, P.conv[0], "image_encoder.features.0.weight", ConvGeom{1, 32, 4, 4, 2, 1, 50, 50, 25, 25, false}, -1, false, true, false);
    build_conv(P, P.conv[1], "image_encoder.features.2.weight", ConvGeom{32, 64, 4, 4, 2, 1, 25, 25, 12, 12, false}, 0, true, false, false);
    build_conv(P, P.conv[2], "image_encoder.features.5.weight", ConvGeom{64, 128, 4, 4, 2, 1, 12, 12, 6, 6, false}, 1, true, false, false);
    build_conv(P, P.conv[3], "image_encoder.features.8.weight", ConvGeom{128, 256, 4, 4, 2, 0, 6, 6, 2, 2, false}, 2, true, false, false);
    // decoder transposed convs (multimnist/model.py:199-208)
    build_conv(P, P.convT[0], "image_decoder.hallucinate.0.weight", ConvGeom{256, 128, 4, 4, 2, 0, 2, 2, 6, 6, true}, 3, true, false, false);
    build_conv(P, P.convT[1], "image_decoder.hallucinate.3.weight", ConvGeom{128, 64, 4, 4, 2, 1, 6, 6, 12, 12, true}, 4, true, false, false);
    build_conv(P, P.convT[2], "image_decoder.hallucinate.6.weight", ConvGeom{64, 32, 5, 5, 2, 1, 12, 12, 25, 25, true}, 5, true, false, false);
    build_conv(P, P.convT[3], "image_decoder.hallucinate.9.weight", ConvGeom{32, 1, 4, 4, 2, 1, 25, 25, 50, 50, true}, -1, true, false, true);
    add_frag_packs(P, P.conv[2]);      // 6x6 / 12x12 layers with 64-256 KB of weights per class: direct-B kernels (convres.hip)
    add_frag_packs(P, P.convT[1]);
    add_frag_packs(P, P.conv[3]);      // the 2x2 <-> 6x6 bottleneck layers (1 MB of weights each)
    add_frag_packs(P, P.convT[0]);

    // classifier (multimnist/model.py:173-179). fc1 consumes the NCHW flatten c*4+y*2+x of a (256,2,2) map that
    // lives here as NHWC [2][2][256]: a 2x2-tap gather with k = (y*2+x)*256 + c.
    {
        LinL& f = P.fc[0];
        f.w_off = off(P, "image_encoder.classifier.0.weight"); f.b_off = off(P, "image_encoder.classifier.0.bias");
        f.N = 400; f.K = 1024; f.pk_dgrad = -1;
        PackDesc d = pack_dense(f.w_off, 400, 1024, npad_for(400), 1024, 1024, 0);
        d.TW = 2; d.C = 256; d.s_ty = 2; d.s_tx = 1; d.s_c = 4;
        f.pk_fwd = P.pk.add(d);
        PackDesc gd = d; gd.Npad = round_up(400, 64);
        f.gk = P.gk.add(gd);
        // dgrad: one class per pixel s of the 2x2 map: da[n][s][c] = sum_j dy[n][j] * W[j][c*4+s]
        for (int sidx = 0; sidx < 4; ++sidx)
            f.pk_dgrad4[sidx] = P.pk.add(pack_dense(f.w_off + sidx, 256, 400, npad_for(256), round_up(400, 64), 4, 1024));
    }
    auto lin = [&](LinL& f, const std::string& n, int N, int K) {
        f.w_off = off(P, n + ".weight"); f.b_off = off(P, n + ".bias"); f.N = N; f.K = K;
        f.pk_fwd = P.pk.add(pack_dense(f.w_off, N, K, npad_for(N), round_up(round_up(K, 8), 64), K, 1));
        f.gk = P.gk.add(pack_dense(f.w_off, N, K, round_up(N, 64), round_up(round_up(K, 8), 64), K, 1));
        f.pk_dgrad = P.pk.add(pack_dense(f.w_off, K, N, npad_for(K), round_up(round_up(N, 8), 64), 1, K));
    };
    lin(P.fc[1], "image_encoder.classifier.3", 200, 400);
    lin(P.fc[2], "image_encoder.classifier.6", 2 * D, 200);
    if (D == 100) {      // the widths mlp_tail.hip is compiled for
        P.mlp_tail = true;
        auto fragd = [&](PackDesc d) { d.frag = 1; return P.pk.add(d); };
        P.mt_w2 = fragd(pack_dense(P.fc[1].w_off, 200, 400, 208, 416, 400, 1));
        P.mt_w3 = fragd(pack_dense(P.fc[2].w_off, 200, 200, 208, 224, 200, 1));
        P.mt_w3t = fragd(pack_dense(P.fc[2].w_off, 200, 200, 208, 224, 1, 200));    // [fc2 unit][fc3 output]
        P.mt_w2t = fragd(pack_dense(P.fc[1].w_off, 400, 200, 400, 224, 1, 400));    // [fc1 unit][fc2 unit]
    }
    {   // upsample Linear(D, 1024): output columns permuted to NHWC n' = s*256 + c  <->  row c*4 + s
        LinL& f = P.up;
        f.w_off = off(P, "image_decoder.upsample.0.weight"); f.b_off = off(P, "image_decoder.upsample.0.bias");
        f.N = 1024; f.K = D;
        // the bias rides in packed column D (the operand z carries a 1.0 there), so its permutation and its
        // gradient (column D of the packed weight gradient) come for free
        PackDesc d = pack_dense(f.w_off, 1024, D, npad_for(1024), round_up(P.ldz, 64), 0, 1);
        d.NL = 256; d.s_nhi = D; d.s_nlo = 4 * D;
        d.bias_off = f.b_off; d.b_nhi = 1; d.b_nlo = 4;
        f.pk_fwd = P.pk.add(d);
        PackDesc gd = d; f.gk = P.gk.add(gd);
        // dgrad: dz[rows][D] = du[rows][1024 NHWC] x W_D[D][1024]
        PackDesc t = pack_dense(f.w_off, D, 1024, npad_for(D), 1024, 1, 0);
        t.TW = 4; t.C = 256; t.s_ty = 0; t.s_tx = D; t.s_c = 4 * D;
        f.pk_dgrad = P.pk.add(t);
    }
    // ---- text packs (row-tile kernels): [round16(N)][round32(K)] bf16, plus transposed copies for backward
    auto rt = [&](long long w, int N, int K, bool transpose) {       // fragment-major: text.hip rowtile_gemm
        PackDesc d = transpose ? pack_dense(w, K, N, round_up(K, 16), round_up(N, 32), 1, K) : pack_dense(w, N, K, round_up(N, 16), round_up(K, 32), K, 1);
        d.frag = 1;
        return P.pk.add(d);
    };
    auto gw = [&](long long w, int N, int K, int Kc) {   // wgrad target: [round64(N)][round64(Kc)] with K valid columns
        return P.gk.add(pack_dense(w, N, K, round_up(N, 64), round_up(Kc, 64), K, 1));
    };
    auto grui = [&](MMPlan::GruIdx& g, const std::string& n, const char* sfx, int inp, int inp_k, bool need_hh) {
        const long long wih = off(P, n + ".weight_ih_" + sfx), whh = off(P, n + ".weight_hh_" + sfx);
        g.bih = off(P, n + ".bias_ih_" + sfx); g.bhh = off(P, n + ".bias_hh_" + sfx);
        g.wih = rt(wih, 300, inp, false); g.wihT = rt(wih, 300, inp, true);
        g.whh = rt(whh, 300, 100, false); g.whhT = rt(whh, 300, 100, true);
        g.g_wih = gw(wih, 300, inp, inp_k);
        g.g_whh = need_hh ? gw(whh, 300, 100, TXT_HP) : -1;
    };
    grui(P.te_f, "text_encoder.gru", "l0", 100, TXT_HP, true);
    grui(P.te_r, "text_encoder.gru", "l0_reverse", 100, TXT_HP, false);
    {
        const long long w = off(P, "text_encoder.h2p.weight");
        P.te_h2p = rt(w, 2 * D, 100, false); P.te_h2pT = rt(w, 2 * D, 100, true); P.g_te_h2p = gw(w, 2 * D, 100, TXT_HP);
    }
    {
        const long long w = off(P, "text_decoder.z2h.weight");
        P.td_z2h = rt(w, 100, D, false); P.td_z2hT = rt(w, 100, D, true); P.g_td_z2h = gw(w, 100, D, P.kz);
    }
    grui(P.td0, "text_decoder.gru", "l0", 100 + D, P.kx, true);
    grui(P.td1, "text_decoder.gru", "l1", 100, TXT_HP, true);
    {
        const long long w = off(P, "text_decoder.h2o.weight");
        P.td_h2o = rt(w, 12, 100 + D, false); P.td_h2oT = rt(w, 12, 100 + D, true); P.g_td_h2o = gw(w, 12, 100 + D, P.kx);
    }
    // Stage of the fused step that refreshes each bf16 copy after an optimizer step (mm_step_body; PackDesc::part of the WEIGHT table):
    //   0  the prologue on the main stream: what the image encoder's FORWARD reads (it starts right behind the prologue)
    //   1  the second-modality stream, in front of the text encoder: every text_encoder.* / text_decoder.* copy
    //   2  the second-modality stream, behind the text encoder's forward: image_decoder.* and the copies only the image encoder's
    //      BACKWARD reads (data-gradient forms) -- the main chain joins that stream before the product of experts, long before it needs them
    // The gather that makes these copies is the slowest part of the prologue (12 of 26 us on the main chain with everything in stage 0).
    for (PackDesc& d : P.pk.d) {
        d.part = 0;
        for (const ParamInfo& pi : P.params)
            if (d.src_off >= pi.offset && d.src_off < pi.offset + pi.numel)
                d.part = pi.name.rfind("text_", 0) == 0 ? 1 : pi.name.rfind("image_decoder.", 0) == 0 ? 2 : 0;
    }
    {
        auto late = [&](int idx) { if (idx >= 0) P.pk.d[idx].part = 2; };
        for (int l = 0; l < 4; ++l)
            for (int i = 0; i < 4; ++i) { late(P.conv[l].pk_dgrad[i]); late(P.conv[l].pk_dgrad_f[i]); }
        for (int i = 0; i < 4; ++i) late(P.fc[0].pk_dgrad4[i]);
        late(P.fc[1].pk_dgrad); late(P.fc[2].pk_dgrad);
        if (P.mlp_tail) { late(P.mt_w3t); late(P.mt_w2t); }
    }
    // gradient descriptors of the decoders (image_decoder.*, text_decoder.*): complete before the encoders' backward has run
    // (MMStepIO::dp_split scatters them into the flat gradient buffer early)
    for (PackDesc& d : P.gk.d)
        for (const ParamInfo& pi : P.params)
            if (d.src_off >= pi.offset && d.src_off < pi.offset + pi.numel)
                d.part = pi.name.rfind("image_decoder.", 0) == 0 || pi.name.rfind("text_decoder.", 0) == 0;
}

// ------------------------------------------------------------------ workspace
void carve(MMPlan& P, Workspace& ws) {
    MMPlan::W& w = P.w;
    const size_t B = P.B, D = P.D, B3 = (size_t)P.carve_passes * B, B2 = (size_t)(P.carve_passes < 2 ? P.carve_passes : 2) * B;
    char* z0 = ws.take<char>(0);
    const int ec[3] = {64, 128, 256}, dc[3] = {128, 64, 32};
    const int SS = MMVAE_STAT_SLOTS;
    for (int i = 0; i < 3; ++i) { w.st_e[i] = ws.take<float2>(SS * ec[i]); w.red_e[i] = ws.take<float2>(SS * ec[i]); }
    for (int i = 0; i < 3; ++i) { w.st_d[i] = ws.take<float2>(3 * SS * dc[i]); w.red_d[i] = ws.take<float2>(3 * SS * dc[i]); }
    w.sums = ws.take<float>(16 * MMVAE_LOSS_SLOTS);
    P.sk_cnt = ws.take<unsigned>(1024);
    w.dz_img = ws.take<float>(B3 * D);
    w.dz_txt = ws.take<float>(B3 * D);
    char* z1 = ws.take<char>(0);
    w.zero_begin = z0; w.zero_bytes = (size_t)(z1 - z0);
    for (int i = 0; i < 3; ++i) { w.aff_e[i] = ws.take<float2>(ec[i]); w.mr_e[i] = ws.take<float2>(ec[i]); }
    for (int i = 0; i < 3; ++i) { w.aff_d[i] = ws.take<float2>(3 * dc[i]); w.mr_d[i] = ws.take<float2>(3 * dc[i]); }
    w.patches1 = ws.take<bf16>(B * 625 * 16);
    w.r1 = ws.take<bf16>(B * 625 * 32); w.r2 = ws.take<bf16>(B * 144 * 64); w.r3 = ws.take<bf16>(B * 36 * 128);
    w.r4 = ws.take<bf16>(B * 4 * 256);
    w.y1 = ws.take<bf16>(B2 * 400); w.y2 = ws.take<bf16>(B2 * 200);
    w.a1 = ws.take<bf16>(B * 625 * 32); w.a2 = ws.take<bf16>(B * 144 * 64); w.a3 = ws.take<bf16>(B * 36 * 128);
    w.a4 = ws.take<bf16>(B * 4 * 256); w.ay1 = ws.take<bf16>(B2 * 400); w.ay2 = ws.take<bf16>(B2 * 200);
    w.au = ws.take<bf16>(B3 * 1024); w.aq1 = ws.take<bf16>(B3 * 36 * 128); w.aq2 = ws.take<bf16>(B3 * 144 * 64);
    w.aq3 = ws.take<bf16>(B3 * 625 * 32);
    w.encout = ws.take<float>(B2 * 2 * D);
    w.m1 = ws.take<uint8_t>(B2 * 400); w.m2 = ws.take<uint8_t>(B2 * 200); w.gkeep = ws.take<uint8_t>(4 * B3 * 100);
    w.txtout = ws.take<float>(B * 2 * D);
    w.te_gates_f = ws.take<float>(4 * 5 * B * 100); w.te_gates_r = ws.take<float>(5 * B * 100);
    w.te_x = ws.take<bf16>(4 * B * 128); w.te_hprev = ws.take<bf16>(4 * B * 128); w.te_hsum = ws.take<bf16>(B * 128);
    w.eps = ws.take<float>(B3 * D); w.mu = ws.take<float>(B3 * D); w.logvar = ws.take<float>(B3 * D);
    w.z_f32 = ws.take<float>(B3 * D); w.z_bf = ws.take<bf16>(B3 * P.ldz);
    w.u = ws.take<bf16>(B3 * 1024); w.q1 = ws.take<bf16>(B3 * 36 * 128); w.q2 = ws.take<bf16>(B3 * 144 * 64);
    w.q3 = ws.take<bf16>(B3 * 625 * 32);
    w.logits = ws.take<float>(B3 * NPIX); w.recon = ws.take<float>(B3 * NPIX); w.dlogit = ws.take<float>(B3 * NPIX);
    w.words = ws.take<float>(B3 * 48); w.dwords = ws.take<float>(B3 * 48); w.tokens = ws.take<long long>(B3 * 4);
    w.td_gates = ws.take<float>(4 * 2 * 5 * B3 * 100);
    w.td_x0 = ws.take<bf16>(4 * B3 * P.kx); w.td_hz = ws.take<bf16>(4 * B3 * P.kx);
    w.td_h0p = ws.take<bf16>(4 * B3 * 128); w.td_mid = ws.take<bf16>(4 * B3 * 128); w.td_h1p = ws.take<bf16>(4 * B3 * 128);
    w.td_zbf = ws.take<bf16>(B3 * P.kz);
    w.dgi0 = ws.take<bf16>(4 * B3 * 304); w.dgh0 = ws.take<bf16>(4 * B3 * 304);
    w.dgi1 = ws.take<bf16>(4 * B3 * 304); w.dgh1 = ws.take<bf16>(4 * B3 * 304);
    w.dlogit_bf = ws.take<bf16>(4 * B3 * 16); w.dhinit = ws.take<bf16>(B3 * 112);
    w.patches4 = ws.take<bf16>(B3 * 625 * 16);
    w.d3 = ws.take<bf16>(B3 * 625 * 32); w.d2 = ws.take<bf16>(B3 * 144 * 64); w.d1 = ws.take<bf16>(B3 * 36 * 128);
    w.du = ws.take<bf16>(B3 * 1024);
    w.d_encout = ws.take<bf16>(B2 * 2 * D); w.d_txtout = ws.take<float>(B * 2 * D);
    w.te_dout_bf = ws.take<bf16>(B * round_up(2 * (int)D, 8));
    w.te_dgi_f = ws.take<bf16>(4 * B * 304); w.te_dgh_f = ws.take<bf16>(4 * B * 304); w.te_dgi_r = ws.take<bf16>(B * 304);
    w.dy2 = ws.take<bf16>(B2 * 200); w.dy1 = ws.take<bf16>(B2 * 400);
    w.db4 = ws.take<bf16>(B2 * 1024); w.dr4 = ws.take<bf16>(B * 1024);
    w.d3e = ws.take<bf16>(B * 36 * 128); w.d2e = ws.take<bf16>(B * 144 * 64); w.d1e = ws.take<bf16>(B * 625 * 32);
    w.d3r = ws.take<bf16>(B3 * 625 * 32); w.d2r = ws.take<bf16>(B3 * 144 * 64); w.d1r = ws.take<bf16>(B3 * 36 * 128);
    w.d3er = ws.take<bf16>(B * 36 * 128); w.d2er = ws.take<bf16>(B * 144 * 64);
    w.tmp_f32 = ws.take<float>(B3 * NPIX);
    P.sk_floats = (size_t)256 * 128 * 128;                 // split-K partial slabs (fully overwritten, never zeroed)
    P.sk_buf = ws.take<float>(P.sk_floats);
    // weight-gradient partial-tile slabs (written and read once per step, never zeroed): gemm.h WgradSlabCtx
    w.slab_floats = (size_t)(P.carve_passes >= 3 ? 48 : 16) << 20;
    w.slab = ws.take<float>(w.slab_floats);
}

// ================================================================== image encoder
// convs once on B images; classifier on variants*B rows (different dropout masks): multimnist/model.py:183-188
// Every layer output is kept twice: raw (r*, y*: backward needs the pre-activation) and activated (a*: the next
// GEMM's operand, staged by pure copies).
// `fuse`: the layers whose consumer is an image-resident kernel (convres.hip) hand it the RAW tensor and the BatchNorm
// statistics (GatherTransform kind 1); the materialised activation is made later, off the main chain, for the weight gradient
int enc_fwd(MMPlan& P, const float* image, int variants, const uint8_t* m1, const uint8_t* m2, int dropout, int training,
            int bn_updates, float* out, hipStream_t s, bool fuse = false) {
    MMPlan::W& w = P.w;
    const int B = P.B;
    if (fuse && mmvae_knob("mm_conv1_mfma", 1)) {      // one workgroup per image on the matrix cores (conv1.hip)
        const PackDesc& d = P.pk.d[P.conv[0].pk_fwd[0]];
        MMVAE_TRY(launch_conv1_fwd_mfma(image, B, P.buf.packed + d.dst_off, d.Kpad, w.r1, w.a1, s));
    } else {
    MMVAE_TRY(launch_im2col_small(image, B, 1, IMG, IMG, 4, 4, 2, 1, 25, 25, w.patches1, 16, s));
    {   // conv1 + Swish (no BatchNorm): the epilogue emits raw and activated
        GatherPlan pl = dense_plan(B * 625, 16, 16, 32);
        GemmParams g = gemm_of(P, pl, P.conv[0].pk_fwd, 1, B * 625);
        g.c.A = w.patches1; g.out_bf = w.r1; g.ldo = 32; g.out_act_bf = w.a1; g.e_act = ACT_SWISH;
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    }
    bf16* r[4] = {w.r1, w.r2, w.r3, w.r4};
    bf16* a[4] = {w.a1, w.a2, w.a3, w.a4};
    for (int l = 1; l < 4; ++l) {
        const ConvL& L = P.conv[l];
        GemmParams g = gemm_of(P, L.fwd, L.pk_fwd, 1, B, L.pk_fwd_f);
        g.c.A = a[l - 1];
        g.out_bf = r[l]; g.ldo = L.g.Cout;
        g.colstats = training ? w.st_e[l - 1] : nullptr;
        GatherTransform tr{};
        if (fuse && l >= 2) {          // conv3 / conv4 stage Swish(BatchNorm(raw output of the layer below)) themselves
            const int prows = B * P.conv[l - 1].g.OH * P.conv[l - 1].g.OW;
            tr.kind = 1;
            tr.fin = bn_fin_args(P, P.bn[P.conv[l - 1].bn], prows, 1, w.st_e[l - 2], bn_updates, w.aff_e[l - 2], w.mr_e[l - 2], training);
            g.c.A = r[l - 1]; g.tr = &tr;
            if (mmvae_knob("mm_stage_out", 1)) tr.out = a[l - 1];      // a2 / a3: operands of the weight gradients, a by-product of the staging
        }
        MMVAE_TRY(launch_gemm_gather(g, s));
        const int rows = B * L.g.OH * L.g.OW;
        if (fuse && l <= 2) continue;  // a2 / a3 are made in front of the next layer's weight gradient (enc_bwd)
        MMVAE_TRY(bn_act(P, P.bn[L.bn], r[l], a[l], rows, rows, 1, w.st_e[l - 1], bn_updates, w.aff_e[l - 1], w.mr_e[l - 1], training, s));
    }
    const int rows = variants * B;
    const bool drop = training && dropout;
    const float ms = 1.f / (1.f - DROP_P);
    {   // fc1 over the NHWC 2x2x256 map (shared by the variants)
        GatherPlan pl = plan_fwdform(2, 2, 1, 1, 256, 2, 2, 1, 0, 400, 1, rows);
        GemmParams g = gemm_of(P, pl, &P.fc[0].pk_fwd, 1, rows);
        g.c.A = w.a4; g.c.a_bcast_n = B;
        g.bias = P.buf.params + P.fc[0].b_off; g.out_bf = w.y1; g.ldo = 400;
        g.out_act_bf = w.ay1; g.e_act = ACT_SWISH; if (drop) { g.e_mask = m1; g.e_mask_scale = ms; }
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    if (P.mlp_tail) {      // classifier.3 + Swish + Dropout + classifier.6 in one row-block launch
        Mlp2FwdArgs a{};
        a.rows = rows; a.x = w.ay1;
        a.w2 = P.buf.packed + P.pk.d[P.mt_w2].dst_off; a.b2 = P.buf.params + P.fc[1].b_off;
        a.w3 = P.buf.packed + P.pk.d[P.mt_w3].dst_off; a.b3 = P.buf.params + P.fc[2].b_off;
        if (drop) { a.mask = m2; a.mask_scale = ms; }
        a.y2 = w.y2; a.ay2 = w.ay2; a.out = out;
        return launch_mlp2_fwd(a, s);
    }
    {
        GatherPlan pl = dense_plan(rows, 400, 400, 200);
        GemmParams g = gemm_of(P, pl, &P.fc[1].pk_fwd, 1, rows);
        g.c.A = w.ay1;
        g.bias = P.buf.params + P.fc[1].b_off; g.out_bf = w.y2; g.ldo = 200;
        g.out_act_bf = w.ay2; g.e_act = ACT_SWISH; if (drop) { g.e_mask = m2; g.e_mask_scale = ms; }
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    {
        GatherPlan pl = dense_plan(rows, 200, 200, 2 * P.D);
        GemmParams g = gemm_of(P, pl, &P.fc[2].pk_fwd, 1, rows);
        g.c.A = w.ay2;
        g.bias = P.buf.params + P.fc[2].b_off; g.out_f = out; g.ldo = 2 * P.D;
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    return MMVAE_OK;
}

// d_out: bf16 [variants*B][2D]; the bias gradient of classifier.6 must already be accumulated by the caller
int enc_bwd(MMPlan& P, const bf16* d_out, int variants, const uint8_t* m1, const uint8_t* m2, int dropout, hipStream_t s, bool fuse = false) {
    MMPlan::W& w = P.w;
    const int B = P.B, rows = variants * B, D2 = 2 * P.D;
    const float ms = 1.f / (1.f - DROP_P);
    // fused step: the three classifier weight gradients (0.04-0.4 GFLOP each, every one a launch at the kernel-latency floor) go
    // out as ONE grouped launch behind conv4's data gradient
    auto cls_w = std::make_shared<std::vector<WgradParams>>();
    if (P.mlp_tail) {      // both data gradients in one row-block launch; the weight gradients follow on the side stream
        {
            GatherPlan pl = dense_plan(rows, 200, 200, D2);
            WgradParams g = wgrad_of(P, pl, &P.fc[2].gk, 1, rows);
            g.c.A = w.ay2; g.P = d_out; g.ldp = D2;
            if (fuse) cls_w->push_back(g);
            else MMVAE_TRY(wgrad_async(P, g, s));
        }
        Mlp2BwdArgs a{};
        a.rows = rows; a.d_out = d_out;
        a.w3t = P.buf.packed + P.pk.d[P.mt_w3t].dst_off; a.w2t = P.buf.packed + P.pk.d[P.mt_w2t].dst_off;
        a.y2 = w.y2; a.y1 = w.y1;
        if (dropout) { a.mask2 = m2; a.mask1 = m1; a.mask_scale = ms; }
        a.dy2 = w.dy2; a.dy1 = w.dy1;
        a.db2 = P.buf.grads + P.fc[1].b_off; a.db1 = P.buf.grads + P.fc[0].b_off;
        MMVAE_TRY(launch_mlp2_bwd(a, s));
        GatherPlan pl = dense_plan(rows, 400, 400, 200);
        WgradParams g = wgrad_of(P, pl, &P.fc[1].gk, 1, rows);
        g.c.A = w.ay1; g.P = w.dy2; g.ldp = 200;
        if (fuse) cls_w->push_back(g);
        else MMVAE_TRY(wgrad_async(P, g, s));
    } else {
    {   // fc3
        GatherPlan pl = dense_plan(rows, 200, 200, D2);
        WgradParams g = wgrad_of(P, pl, &P.fc[2].gk, 1, rows);
        g.c.A = w.ay2; g.P = d_out; g.ldp = D2;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, D2, D2, 200);
        GemmParams d = gemm_of(P, pd, &P.fc[2].pk_dgrad, 1, rows);
        d.c.A = d_out; d.out_bf = w.dy2; d.ldo = 200;
        d.d_r = w.y2; d.d_ld = 200; d.d_act = ACT_SWISH; if (dropout) { d.d_mask = m2; d.d_mask_scale = ms; }
        d.d_colsum = P.buf.grads + P.fc[1].b_off;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    {   // fc2
        GatherPlan pl = dense_plan(rows, 400, 400, 200);
        WgradParams g = wgrad_of(P, pl, &P.fc[1].gk, 1, rows);
        g.c.A = w.ay1; g.P = w.dy2; g.ldp = 200;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, 200, 200, 400);
        GemmParams d = gemm_of(P, pd, &P.fc[1].pk_dgrad, 1, rows);
        d.c.A = w.dy2; d.out_bf = w.dy1; d.ldo = 400;
        d.d_r = w.y1; d.d_ld = 400; d.d_act = ACT_SWISH; if (dropout) { d.d_mask = m1; d.d_mask_scale = ms; }
        d.d_colsum = P.buf.grads + P.fc[0].b_off;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    }
    {   // fc1: wgrad gathers the shared 2x2x256 map; dgrad emits NHWC gradients for `rows` samples
        GatherPlan pl = plan_fwdform(2, 2, 1, 1, 256, 2, 2, 1, 0, 400, 1, rows);
        WgradParams g = wgrad_of(P, pl, &P.fc[0].gk, 1, rows);
        g.c.A = w.a4; g.c.a_bcast_n = B;
        g.P = w.dy1; g.ldp = 400;
        if (P.mlp_tail && fuse) {
            MMPlan* pp = &P;
            cls_w->push_back(g);
            side_later(P, [pp, cls_w](hipStream_t ws_) {
                MMVAE_TRY(launch_wgrad_group(cls_w->data(), (int)cls_w->size(), ws_, &pp->slab));
                return pp->batch_reduce ? MMVAE_OK : launch_wgrad_reduce(&pp->slab, ws_, true);
            }, mmvae_knob("mm_cls_lane", 1));
        } else {
            MMVAE_TRY(wgrad_async(P, g, s));
        }
        // dgrad: 4 classes = the 4 pixels of the 2x2 map, each with its own [256][400] matrix
        GatherPlan pd{};
        pd.c.AH = 1; pd.c.AW = 1; pd.c.Ald = 400; pd.c.C = 400; pd.c.sy = pd.c.sx = 1; pd.c.dy = pd.c.dx = 1;
        pd.c.OH = 2; pd.c.OW = 2; pd.c.osy = pd.c.osx = 1; pd.c.N = 256; pd.c.nclasses = 4;
        for (int sidx = 0; sidx < 4; ++sidx) {
            GatherClass& k = pd.cls[sidx];
            k.OY = 1; k.OX = 1; k.TH = 1; k.TW = 1; k.offy = 0; k.offx = 0; k.ooy = sidx / 2; k.oox = sidx % 2;
            k.K = 400; k.Kpad = round_up(400, 64);
        }
        GemmParams d = gemm_of(P, pd, P.fc[0].pk_dgrad4, 1, rows);
        d.c.A = w.dy1; d.out_bf = w.db4; d.ldo = 256;
        d.d_r = w.r4; d.d_ld = 256; d.d_bcast_n = B; d.d_act = ACT_SWISH;
        d.d_affine = w.aff_e[2]; d.d_meanrstd = w.mr_e[2]; d.d_red = w.red_e[2];
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    // ---- conv stack (features shared by all variants: gradients of the variants add up)
    bf16* r[4] = {w.r1, w.r2, w.r3, w.r4};
    bf16* a[4] = {w.a1, w.a2, w.a3, w.a4};
    bf16* dr[4] = {w.d1e, w.d2e, w.d3e, w.dr4};
    bf16* drr[4] = {nullptr, w.d2er, w.d3er, nullptr};       // fused layers: BatchNorm-backward output apart from db
    for (int l = 3; l >= 1; --l) {
        const ConvL& L = P.conv[l];
        const BnL& b = P.bn[L.bn];
        const int pix = L.g.OH * L.g.OW;
        // conv3 / conv2: the image-resident data-gradient kernel applies the BatchNorm backward to db while staging it
        // (GatherTransform kind 2); the materialised dr is only the weight gradient's operand and is made in front of it,
        // off the main chain, into a buffer of its own (db is still being read by the data gradient)
        const bool fl = fuse && l < 3;
        BnBwdApplyArgs x{};
        x.db = (l == 3) ? w.db4 : dr[l];
        x.db2 = (l == 3 && variants == 2) ? w.db4 + (size_t)B * 1024 : nullptr;
        x.r = r[l]; x.dr = fl ? drr[l] : dr[l]; x.rows = B * pix; x.C = L.g.Cout; x.ld = L.g.Cout; x.rows_per_group = B * pix; x.G = 1;
        x.red = w.red_e[l - 1]; x.meanrstd = w.mr_e[l - 1]; x.gamma = P.buf.params + b.w_off;
        x.dgamma = P.buf.grads + b.w_off; x.dbeta = P.buf.grads + b.b_off;
        if (!fl) MMVAE_TRY(launch_bn_bwd_apply(x, s));
        // wgrad: P = dr[l] (rows over the output grid), G = activated input gathered in forward form.  Forks are bound to a
        // producer kernel's completion: a fused layer's operands came out of the previous data gradient (the current event);
        // the unfused conv4 issues its weight gradient behind its own data gradient, on the event conv3's needs anyway
        WgradParams gw = wgrad_of(P, L.fwd, L.gk, 1, B);
        gw.c.A = a[l - 1]; gw.P = x.dr; gw.ldp = L.g.Cout;
        if (fuse) {
            // fused layers (conv3, conv2): the weight gradient runs on a side stream on operands that are by-products of the main
            // chain's staging (GatherTransform::out) or are materialised right in front of it, off the main chain
            MMPlan* pp = &P;
            {
                // With staging by-products the forward of conv3 / conv4 left a2 / a3 behind and conv3's data gradient below
                // leaves its BatchNorm-backward dr behind: nothing to materialise in front of the kernel
                const bool byp = mmvae_knob("mm_stage_out", 1) != 0;
                const bool actp = l >= 2 && !byp;      // a[l-1] = Swish(BatchNorm(r[l-1])) was never materialised (conv3, conv4)
                const bool bnb = fl && !(byp && l == 2);
                const int prows = B * P.conv[l >= 2 ? l - 1 : 1].g.OH * P.conv[l >= 2 ? l - 1 : 1].g.OW;
                const BnL bp = P.bn[P.conv[l >= 2 ? l - 1 : 1].bn];
                const bf16* rin = r[l >= 2 ? l - 1 : 1]; bf16* ao = a[l >= 2 ? l - 1 : 1]; const float2* st = w.st_e[l >= 2 ? l - 2 : 0];
                WgradParams g0 = wgrad_of(P, L.fwd, L.gk, 1, B);
                g0.c.A = a[l - 1]; g0.P = x.dr; g0.ldp = L.g.Cout;
                // lane 1: the LAST weight gradients of the step (conv3's, conv2's) go to the second-modality stream -- it has run dry
                // by then, and the weight-gradient stream still holds the classifier's and conv4's
                side_later(P, [pp, x, g0, bnb, actp, bp, rin, ao, st, prows](hipStream_t ws_) {
                    if (bnb) MMVAE_TRY(launch_bn_bwd_apply(x, ws_));
                    if (actp) MMVAE_TRY(bn_act_side(*pp, bp, rin, ao, prows, prows, 1, st, 1, ws_));
                    return wgrad_on(*pp, g0, ws_);
                }, l == 1 ? 1 : l == 2 ? mmvae_knob("mm_conv3_lane", 0) : mmvae_knob("mm_conv4_lane", 0));
            }
            if (fuse && l == 1) MMVAE_TRY(side_flush(P, s));    // conv2's: its operands came out of conv3's data gradient (current event)
        }
        {   // dgrad (class form) with the d-activation of the producer layer fused in the epilogue
            GemmParams d = gemm_of(P, L.dgrad, L.pk_dgrad, 1, B, L.pk_dgrad_f);
            d.c.A = dr[l]; d.out_bf = dr[l - 1]; d.ldo = L.g.Cin;
            d.d_r = r[l - 1]; d.d_ld = L.g.Cin; d.d_act = ACT_SWISH;
            if (l > 1) { d.d_affine = w.aff_e[l - 2]; d.d_meanrstd = w.mr_e[l - 2]; d.d_red = w.red_e[l - 2]; }
            GatherTransform tr{};
            if (fl) {
                tr.kind = 2; tr.r = r[l]; tr.red = w.red_e[l - 1]; tr.mr = w.mr_e[l - 1]; tr.gamma = P.buf.params + b.w_off;
                tr.inv_cnt = 1.f / (float)(B * pix); tr.groups = 1;
                if (l == 2 && mmvae_knob("mm_stage_out", 1)) {      // conv3: dr for its weight gradient + the BatchNorm parameter gradients
                    tr.out = drr[l]; tr.dgamma = P.buf.grads + b.w_off; tr.dbeta = P.buf.grads + b.b_off;
                }
                d.tr = &tr;
            }
            // forks: behind conv4's data gradient (classifier + conv4 + conv3 weight gradients) and behind conv3's (conv2's)
            // (knob mm_forks = 1: ONE fork for the whole encoder backward, behind conv3's data gradient -- a kernel that carries a
            //  completion event ends with a system-scope release, ~8 us on the main chain per fork: tools/step_parts.py)
            const bool flush = fuse && P.wgrad_forked && (mmvae_knob("mm_forks", 0) ? l == 2 : l >= 2);
            if (flush) arm_fork(P);
            MMVAE_TRY(launch_gemm_gather(d, s));
            if (flush) { MMVAE_TRY(commit_fork(P, s)); MMVAE_TRY(side_flush(P, s)); }
        }
        if (!fuse) MMVAE_TRY(wgrad_async(P, gw, s));
    }
    if (fuse && !P.wgrad_forked) MMVAE_TRY(side_flush(P, s));     // unforked (serial) use: nothing to wait for
    if (fuse && mmvae_knob("mm_conv1_mfma", 1) && P.step_image) {       // the LAST kernel of the backward chain: conv1.hip
        const PackDesc& gd = P.gk.d[P.conv[0].gk[0]];
        MMVAE_TRY(launch_conv1_wgrad_mfma(P.step_image, B, w.d1e, P.buf.gpk + gd.dst_off, gd.Kpad, s));
    } else {   // conv1 wgrad over the im2col patches: the LAST kernel of the backward chain -- it stays on the main stream (a hop
        // to a side stream and back would put two event latencies on the critical path)
        GatherPlan pl = dense_plan(B * 625, 16, 16, 32);
        WgradParams g = wgrad_of(P, pl, P.conv[0].gk, 1, B * 625);
        g.c.A = w.patches1; g.P = w.d1e; g.ldp = 32;
        // fp32 atomics into the (zeroed) packed gradient instead of slab copies + a reduce launch: this is the last kernel in
        // front of the optimizer, a second launch here is pure tail latency (the tile is 32 x 16)
        MMVAE_TRY(launch_wgrad(g, s, nullptr));
    }
    return MMVAE_OK;
}

// ================================================================== image decoder (multimnist/model.py:211-216)
// z_bf: [groups*B][ldz] with column D == 1.0 (folded bias).  The last layer is the fused direct kernel
// ConvTranspose2d(32,1) + sigmoid (+ BCE and its gradient when `bce` is given).
// BatchNorm finalize of hallucinate.7 + the thin last layer + (for the first bwd_groups groups) its gradients: thin.h
int fused_tail(MMPlan& P, int groups, int training, const ConvTLastFwdArgs* last, int last_groups, int fused_bwd_groups, hipStream_t s) {
    MMPlan::W& w = P.w;
    const int B = P.B;
    const ConvL& L = P.convT[2];
    const BnL& b = P.bn[L.bn];
    DecLastFusedArgs x{};
    x.r = w.q3; x.act = ACT_SWISH; x.w = P.buf.params + P.convT[3].w_off;
    x.G = last_groups > 0 ? last_groups : groups; x.B = B; x.IH = 25; x.IW = 25; x.Cin = 32;
    x.bwd_groups = std::min(fused_bwd_groups, x.G);
    BnFinalizeArgs& f = x.fin;
    f.stats = w.st_d[2]; f.G = groups; f.C = b.C; f.count = (float)(B * L.g.OH * L.g.OW);
    f.gamma = P.buf.params + b.w_off; f.beta = P.buf.params + b.b_off;
    f.running_mean = P.buf.bn_stats + b.stat_off; f.running_var = P.buf.bn_stats + b.stat_off + b.C;
    f.num_batches_tracked = P.buf.bn_nbt + b.idx;
    f.updates_per_group = 1; f.affine = w.aff_d[2]; f.meanrstd = w.mr_d[2]; f.eps = BN_EPS; f.momentum = BN_MOM; f.training = training;
    f.skip_update_mask = groups > 1 ? P.dec_skip_mask : 0u;
    x.target = last->target; x.logits = last->logits; x.recon = last->recon; x.dlogit = nullptr;
    for (int k = 0; k < 4; ++k) x.coef[k] = last->coef[k];
    x.loss_sum = last->loss_sum;
    if (x.bwd_groups > 0) {
        const int chunks = x.bwd_groups * B * (dec_last_mfma_applies(x) ? 1 : dec_last_fused_strips(25));
        x.wslab = P.slab.take((size_t)chunks * 32 * 16);
        MMVAE_REQUIRE(x.wslab != nullptr, "fused decoder tail: the weight-gradient slab pool is exhausted");
        x.db = w.d3; x.red = w.red_d[2];
        WgradSlabJob j{};
        const PackDesc& gd = P.gk.d[P.convT[3].gk[0]];
        j.dst = P.buf.gpk + gd.dst_off; j.slab = x.wslab; j.N = 32; j.K = 16; j.Kpad = gd.Kpad; j.chunks = chunks;
        j.chunk_stride = 32 * 16; j.src_ld = 16; j.stream = s;
        P.slab.jobs.push_back(j);
    }
    return launch_dec_last_fused(x, s);
}

// `fused_bwd_groups` >= 0: the fused tail (thin.h DecLastFusedArgs) replaces bn_act of the last BatchNorm + the thin last
// layer, and for the first fused_bwd_groups groups also the last layer's data / weight gradients (step path only: the
// granular modules get their upstream gradient from autograd and keep the separate kernels).
int dec_fwd(MMPlan& P, int groups, int training, ConvTLastFwdArgs* last, hipStream_t s, int last_groups = -1, int fused_bwd_groups = -1,
            bool fuse = false) {
    MMPlan::W& w = P.w;
    const int B = P.B, rows = groups * B;
    {
        GatherPlan pl = dense_plan(rows, P.ldz, P.ldz, 1024);
        GemmParams g = gemm_of(P, pl, &P.up.pk_fwd, 1, rows);
        g.c.A = w.z_bf; g.out_bf = w.u; g.ldo = 1024; g.out_act_bf = w.au; g.e_act = ACT_SWISH;
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    bf16* q[4] = {w.u, w.q1, w.q2, w.q3};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    for (int l = 0; l < 3; ++l) {
        const ConvL& L = P.convT[l];
        {
            GemmParams g = gemm_of(P, L.fwd, L.pk_fwd, groups, B, L.pk_fwd_f);
            g.c.A = aq[l];
            g.out_bf = q[l + 1]; g.ldo = L.g.Cout;
            g.colstats = training ? w.st_d[l] : nullptr;
            GatherTransform tr{};
            if (fuse && l >= 1) {      // convT2 / convT3 stage Swish(BatchNorm(q[l])) themselves
                const ConvL& Lp = P.convT[l - 1];
                const int prpg = B * Lp.g.OH * Lp.g.OW;
                tr.kind = 1;
                tr.fin = bn_fin_args(P, P.bn[Lp.bn], prpg, groups, w.st_d[l - 1], 1, w.aff_d[l - 1], w.mr_d[l - 1], training);
                g.c.A = q[l]; g.tr = &tr;
                if (mmvae_knob("mm_stage_out", 1)) tr.out = aq[l];     // operand of this layer's weight gradient, a by-product of the staging
            }
            MMVAE_TRY(launch_gemm_gather(g, s));
        }
        const int rpg = B * L.g.OH * L.g.OW;
        if (l == 2 && fused_bwd_groups >= 0) continue;
        if (fuse && l <= 1) continue;  // aq[l+1] is made in front of the next layer's weight gradient (dec_bwd)
        MMVAE_TRY(bn_act(P, P.bn[L.bn], q[l + 1], aq[l + 1], groups * rpg, rpg, groups, w.st_d[l], 1, w.aff_d[l], w.mr_d[l], training, s));
    }
    if (fused_bwd_groups >= 0) return fused_tail(P, groups, training, last, last_groups, fused_bwd_groups, s);
    ConvTLastFwdArgs x = *last;
    x.act = w.aq3; x.w = P.buf.params + P.convT[3].w_off; x.G = last_groups > 0 ? last_groups : groups; x.B = B; x.IH = 25; x.IW = 25; x.Cin = 32; x.Cout = 1;
    return launch_convt_last_fwd(x, s);
}

// dlogit: fp32 NCHW [groups*B][1][50][50] (grad wrt the pre-sigmoid logits). Writes dz (fp32 [groups*B][D]).
int dec_bwd(MMPlan& P, const float* dlogit, int groups, float* dz, hipStream_t s, bool last_fused = false, bool fuse = false,
            int fwd_groups = 0, int training = 1) {
    MMPlan::W& w = P.w;
    const int B = P.B, rows = groups * B;
    bf16* q[4] = {w.u, w.q1, w.q2, w.q3};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    bf16* dq[4] = {w.du, w.d1, w.d2, w.d3};
    bf16* dqr[4] = {nullptr, w.d1r, w.d2r, w.d3r};         // fused layers: BatchNorm-backward output apart from db
    if (last_fused) {
        // d3, the BatchNorm-backward sums and the weight-gradient partials of the last layer came out of the fused tail
        // (dec_fwd); only the sum of the partials is left, off the main chain
        if (P.wgrad_forked) {
            PlanBase* pb = &P;
            side_later(P, [pb, s](hipStream_t wst) {
                for (WgradSlabJob& j : pb->slab.jobs) if (j.stream == s && j.src_ld == 16) j.stream = wst;
                return launch_wgrad_reduce(&pb->slab, wst, true);
            });
        }
    } else {   // last transposed conv (32 -> 1): both gradients go through the im2col patches of dlogit (K = 16 taps):
        // input gradient = dense GEMM patches x W (d-Swish + BatchNorm-backward sums in the epilogue), weight
        // gradient = patches^T x activated input.  (The direct dot-product kernel convt_last_dgrad is VALU-bound:
        // 52 us against 39 us for this GEMM at B=256.)
        const ConvL& L = P.convT[3];
        MMVAE_TRY(launch_im2col_small(dlogit, rows, 1, IMG, IMG, 4, 4, 2, 1, 25, 25, w.patches4, 16, s));
        {
            GatherPlan pd = dense_plan(B * 625, 16, 16, 32);
            GemmParams d = gemm_of(P, pd, L.pk_dgrad, groups, B * 625);
            d.c.A = w.patches4; d.out_bf = w.d3; d.ldo = 32;
            d.d_r = w.q3; d.d_ld = 32; d.d_act = ACT_SWISH; d.d_affine = w.aff_d[2]; d.d_meanrstd = w.mr_d[2]; d.d_red = w.red_d[2];
            arm_fork(P);
            MMVAE_TRY(launch_gemm_gather(d, s));
            MMVAE_TRY(commit_fork(P, s));
        }
        GatherPlan pl = plan_fwdform(1, 1, 25, 25, 16, 1, 1, 1, 0, 32, groups, B);   // rows (n, iy, ix), dense K=16
        WgradParams g = wgrad_of(P, pl, L.gk, groups, B);
        g.c.A = w.patches4; g.c.AH = 25; g.c.AW = 25; g.c.sy = g.c.sx = 1;
        g.P = w.aq3; g.ldp = 32;
        MMVAE_TRY(wgrad_async(P, g, s));
    }
    for (int l = 2; l >= 0; --l) {
        const ConvL& L = P.convT[l];
        const BnL& b = P.bn[L.bn];
        const int pix = L.g.OH * L.g.OW;
        // convT3 / convT2: the image-resident data-gradient kernel applies the BatchNorm backward while it stages db
        // (GatherTransform kind 2); dr and the activated layer input are only the weight gradient's operands and are made in
        // front of it on the side stream (enc_bwd has the same arrangement)
        const bool fl = fuse;          // (hallucinate.0 too: its input is the upsample Linear's activation, no BatchNorm below it)
        BnBwdApplyArgs x{};
        x.db = dq[l + 1]; x.r = q[l + 1]; x.dr = fl ? dqr[l + 1] : dq[l + 1]; x.rows = rows * pix; x.C = L.g.Cout; x.ld = L.g.Cout;
        x.rows_per_group = B * pix; x.G = groups;
        x.red = w.red_d[l]; x.meanrstd = w.mr_d[l]; x.gamma = P.buf.params + b.w_off;
        x.dgamma = P.buf.grads + b.w_off; x.dbeta = P.buf.grads + b.b_off;
        if (!fl) MMVAE_TRY(launch_bn_bwd_apply(x, s));
        // Forks are bound to a producer kernel's completion (arm_fork / commit_fork).  Fused layers: the operands of the
        // weight gradient came out of the previous data gradient (or the fused tail), whose event is the current one.  The
        // unfused first layer: its weight gradient is issued behind the layer's own data gradient, on the event the upsample
        // weight gradient needs anyway (one bound event less on the main chain)
        WgradParams gw = convT_wgrad(P, L, groups, B, aq[l], x.dr);
        if (fl) {
            MMPlan* pp = &P;
            const ConvL& Lp = P.convT[l >= 1 ? l - 1 : 0];
            const int prpg = B * Lp.g.OH * Lp.g.OW;
            {
                // the streamed weight-gradient kernel on operands materialised right in front of it, on the side stream: the
                // BatchNorm backward of db (its own buffer: the data gradient on the main chain still reads db) and, above the
                // first layer, the layer input aq[l] = Swish(BatchNorm(q[l])) that the forward never wrote
                // (with staging by-products, GatherTransform::out, both come out of kernels of the main chain instead: aq[l]
                //  out of this layer's forward, dr out of this layer's data gradient below -- the side work is the GEMM alone)
                const bool byp = mmvae_knob("mm_stage_out", 1) != 0;
                const bool actp = l >= 1 && !byp;
                const BnL bp = P.bn[Lp.bn];
                const bf16* qin = q[l]; bf16* aqo = aq[l]; const float2* st = w.st_d[l >= 1 ? l - 1 : 0];
                side_later(P, [pp, x, gw, byp, actp, bp, qin, aqo, st, groups, prpg, training](hipStream_t ws_) {
                    if (!byp) MMVAE_TRY(launch_bn_bwd_apply(x, ws_));
                    if (actp) MMVAE_TRY(bn_act_side(*pp, bp, qin, aqo, groups * prpg, prpg, groups, st, training, ws_));
                    return wgrad_on(*pp, gw, ws_);
                });
            }
        } else if (fuse || l == 0) {
            MMPlan* pp = &P;
            side_later(P, [pp, gw](hipStream_t ws_) { return wgrad_on(*pp, gw, ws_); });
        }
        {
            GemmParams d = gemm_of(P, L.dgrad, L.pk_dgrad, groups, B, L.pk_dgrad_f);
            d.c.A = dq[l + 1]; d.out_bf = dq[l]; d.ldo = L.g.Cin;
            d.d_r = q[l]; d.d_ld = L.g.Cin; d.d_act = ACT_SWISH;
            if (l > 0) { d.d_affine = w.aff_d[l - 1]; d.d_meanrstd = w.mr_d[l - 1]; d.d_red = w.red_d[l - 1]; }
            GatherTransform tr{};
            if (fl) {
                tr.kind = 2; tr.r = q[l + 1]; tr.red = w.red_d[l]; tr.mr = w.mr_d[l]; tr.gamma = P.buf.params + b.w_off;
                tr.inv_cnt = 1.f / (float)(B * pix); tr.groups = groups;
                if (mmvae_knob("mm_stage_out", 1)) { tr.out = dqr[l + 1]; tr.dgamma = P.buf.grads + b.w_off; tr.dbeta = P.buf.grads + b.b_off; }
                d.tr = &tr;
            }
            // the side work collected so far forks off the completion of the first and of the last data gradient
            const bool flush = P.wgrad_forked && ((l == 2 && !mmvae_knob("mm_forks", 0)) || l == 0);
            if (flush) arm_fork(P);
            MMVAE_TRY(launch_gemm_gather(d, s));
            if (flush) { MMVAE_TRY(commit_fork(P, s)); if (l == 2) MMVAE_TRY(side_flush(P, s)); }
        }
        if (!fl && !(fuse || l == 0)) MMVAE_TRY(wgrad_async(P, gw, s));
    }
    {   // upsample Linear: weight (+ folded bias) gradient and dz
        GatherPlan pl = dense_plan(rows, P.ldz, P.ldz, 1024);
        WgradParams g = wgrad_of(P, pl, &P.up.gk, 1, rows);
        g.c.A = w.z_bf; g.P = w.du; g.ldp = 1024;
        MMPlan* pp = &P;
        side_later(P, [pp, g](hipStream_t ws_) { return wgrad_on(*pp, g, ws_); });
        MMVAE_TRY(side_flush(P, s));                // forks off the last data gradient above (du is its output)
        GatherPlan pd = dense_plan(rows, 1024, 1024, P.D);
        GemmParams d = gemm_of(P, pd, &P.up.pk_dgrad, 1, rows);
        d.c.A = w.du; d.out_f = dz; d.ldo = P.D;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    return MMVAE_OK;
}

// ================================================================== text modules
GruPacked gru_of(const MMPlan& P, const MMPlan::GruIdx& g) {
    GruPacked r{};
    r.wih = P.buf.packed + P.pk.d[g.wih].dst_off; r.kih = P.pk.d[g.wih].Kpad;
    r.whh = P.buf.packed + P.pk.d[g.whh].dst_off;
    r.wihT = P.buf.packed + P.pk.d[g.wihT].dst_off; r.nih = P.pk.d[g.wihT].Npad;
    r.whhT = P.buf.packed + P.pk.d[g.whhT].dst_off;
    r.bih = P.buf.params + g.bih; r.bhh = P.buf.params + g.bhh;
    return r;
}
TextEncArgs te_args(MMPlan& P, const long long* text, float* out, bool save) {
    MMPlan::W& w = P.w;
    TextEncArgs a{};
    a.B = P.B; a.D = P.D; a.tokens = text; a.embed = P.buf.params + off(P, "text_encoder.embed.weight");
    a.fwd = gru_of(P, P.te_f); a.rev = gru_of(P, P.te_r);
    a.h2p = P.buf.packed + P.pk.d[P.te_h2p].dst_off; a.nh2p = P.pk.d[P.te_h2p].Npad;
    a.h2pT = P.buf.packed + P.pk.d[P.te_h2pT].dst_off;
    a.h2p_bias = P.buf.params + off(P, "text_encoder.h2p.bias");
    a.out = out;
    if (save) { a.gates_f = w.te_gates_f; a.gates_r = w.te_gates_r; a.x_bf = w.te_x; a.hprev_bf = w.te_hprev; a.hsum_bf = w.te_hsum; }
    return a;
}
int txt_enc_bwd(MMPlan& P, const long long* text, const float* d_out, hipStream_t s) {
    MMPlan::W& w = P.w;
    const int B = P.B, D2 = 2 * P.D;
    TextEncBwdArgs a{};
    a.f = te_args(P, text, nullptr, true);
    a.d_out = d_out; a.d_out_bf = w.te_dout_bf;
    a.dgi_f = w.te_dgi_f; a.dgh_f = w.te_dgh_f; a.dgi_r = w.te_dgi_r;
    float* G = P.buf.grads;
    a.g_embed = G + off(P, "text_encoder.embed.weight");
    a.g_bih_f = G + P.te_f.bih; a.g_bhh_f = G + P.te_f.bhh; a.g_bih_r = G + P.te_r.bih; a.g_bhh_r = G + P.te_r.bhh;
    a.g_h2p_bias = G + off(P, "text_encoder.h2p.bias");
    MMVAE_TRY(launch_text_encoder_bwd(a, s));
    // the four weight gradients of the text encoder share ONE grouped launch
    WgradParams list[4];
    int nl = 0;
    auto wg = [&](int gidx, const bf16* Pm, int N, int ldp, const bf16* Gm, int C, int rows) {
        GatherPlan pl = dense_plan(rows, C, C, N);
        WgradParams g = wgrad_of(P, pl, &gidx, 1, rows);
        g.c.A = Gm; g.P = Pm; g.ldp = ldp;
        list[nl++] = g;
    };
    wg(P.te_f.g_wih, w.te_dgi_f, 300, 304, w.te_x, TXT_HP, 4 * B);
    wg(P.te_f.g_whh, w.te_dgh_f, 300, 304, w.te_hprev, TXT_HP, 4 * B);
    wg(P.te_r.g_wih, w.te_dgi_r, 300, 304, w.te_x + (size_t)3 * B * TXT_HP, TXT_HP, B);
    wg(P.g_te_h2p, w.te_dout_bf, D2, round_up(D2, 8), w.te_hsum, TXT_HP, B);
    MMVAE_TRY(launch_wgrad_group(list, nl, s, &P.slab));
    return launch_wgrad_reduce(&P.slab, s, true);
}
TextDecArgs td_args(MMPlan& P, const float* z, int groups, bool save) {
    MMPlan::W& w = P.w;
    TextDecArgs a{};
    a.R = groups * P.B; a.D = P.D; a.rows_per_pass = P.B; a.z = z;
    a.embed = P.buf.params + off(P, "text_decoder.embed.weight");
    a.z2h = P.buf.packed + P.pk.d[P.td_z2h].dst_off; a.kz = P.kz;
    a.z2hT = P.buf.packed + P.pk.d[P.td_z2hT].dst_off;
    a.z2h_bias = P.buf.params + off(P, "text_decoder.z2h.bias");
    a.l0 = gru_of(P, P.td0); a.l1 = gru_of(P, P.td1);
    a.h2o = P.buf.packed + P.pk.d[P.td_h2o].dst_off; a.kx = P.kx;
    a.h2oT = P.buf.packed + P.pk.d[P.td_h2oT].dst_off;
    a.h2o_bias = P.buf.params + off(P, "text_decoder.h2o.bias");
    a.words = w.words; a.tokens_out = w.tokens;
    if (save) {
        a.gates = w.td_gates; a.x0_bf = w.td_x0; a.h0p_bf = w.td_h0p; a.mid_bf = w.td_mid; a.h1p_bf = w.td_h1p;
        a.hz_bf = w.td_hz; a.z_bf = w.td_zbf;
    }
    return a;
}
int txt_dec_bwd(MMPlan& P, const TextDecArgs& f, const float* dwords, float* dz, hipStream_t s) {
    MMPlan::W& w = P.w;
    const int R = f.R, XI = 100 + P.D;
    TextDecBwdArgs a{};
    a.f = f; a.dwords = dwords; a.R_active = R; a.dz = dz;
    a.dgi0 = w.dgi0; a.dgh0 = w.dgh0; a.dgi1 = w.dgi1; a.dgh1 = w.dgh1; a.dlogit_bf = w.dlogit_bf; a.dhinit_bf = w.dhinit;
    float* G = P.buf.grads;
    a.g_embed = G + off(P, "text_decoder.embed.weight");
    a.g_b[0] = G + P.td0.bih; a.g_b[1] = G + P.td0.bhh; a.g_b[2] = G + P.td1.bih; a.g_b[3] = G + P.td1.bhh;
    a.g_h2o_bias = G + off(P, "text_decoder.h2o.bias"); a.g_z2h_bias = G + off(P, "text_decoder.z2h.bias");
    // In the fused step the main chain needs dz (and the NLL sums) of THIS kernel, not the weight gradients enqueued behind it on the
    // same stream: the join is the kernel's own completion event (a join on "everything enqueued on T so far" made the main chain wait
    // for ~45 us of weight-gradient launches; measured on the traces of round 4: the second-modality stream was co-critical)
    P.ev_dz = nullptr;
    const bool own_ev = P.in_step && s == P.st_text && mmvae_knob("mm_dz_event", 1) != 0;
    if (own_ev) {
        P.ev_dz = next_ev(P);
        if (!P.capturing && !mmvae_knob("no_stop_events", 0)) mmvae_arm_stop_event(P.ev_dz);
    }
    MMVAE_TRY(launch_text_decoder_bwd(a, s));
    if (own_ev && (mmvae_take_stop_event() != nullptr || P.capturing || mmvae_knob("no_stop_events", 0)) && hipEventRecord(P.ev_dz, s) != hipSuccess) {
        mmvae_set_error("text decoder event failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    // the six weight gradients of the text decoder: one grouped launch (128x128 tile class) + the thin h2o one
    WgradParams list[6];
    int nl = 0;
    auto wg = [&](int gidx, const bf16* Pm, int N, int ldp, const bf16* Gm, int C, int rows) {
        GatherPlan pl = dense_plan(rows, C, C, N);
        WgradParams g = wgrad_of(P, pl, &gidx, 1, rows);
        g.c.A = Gm; g.P = Pm; g.ldp = ldp;
        list[nl++] = g;
    };
    (void)XI;
    wg(P.td0.g_wih, w.dgi0, 300, 304, w.td_x0, P.kx, 4 * R);
    wg(P.td0.g_whh, w.dgh0, 300, 304, w.td_h0p, TXT_HP, 4 * R);
    wg(P.td1.g_wih, w.dgi1, 300, 304, w.td_mid, TXT_HP, 4 * R);
    wg(P.td1.g_whh, w.dgh1, 300, 304, w.td_h1p, TXT_HP, 4 * R);
    wg(P.g_td_z2h, w.dhinit, 100, 112, w.td_zbf, P.kz, R);
    wg(P.g_td_h2o, w.dlogit_bf, 12, 16, w.td_hz, P.kx, 4 * R);
    MMVAE_TRY(launch_wgrad_group(list, nl, s, &P.slab));
    MMVAE_TRY(launch_wgrad_reduce(&P.slab, s, true));
    // every gradient of text_decoder.* is complete on this stream here: the early optimizer part waits for this event
    P.ev_txtgrads = nullptr;
    if (own_ev) {
        P.ev_txtgrads = next_ev(P);
        if (hipEventRecord(P.ev_txtgrads, s) != hipSuccess) {
            mmvae_set_error("text decoder event failed: %s", hipGetErrorString(hipGetLastError()));
            return MMVAE_EHIP;
        }
    }
    return MMVAE_OK;
}

}  // namespace

// ================================================================== public (internal C++) API
MMPlan* mm_create(int D, int B) {
    if (D < 1 || D > 127 || B < 1) { mmvae_set_error("mm_create: need 1 <= n_latents <= 127 and batch >= 1"); return nullptr; }
    MMPlan* P = new MMPlan();
    P->D = D; P->B = B;
    // round 3: two weight-gradient streams side by side (the chain was the step's tail: 758 -> 718 us); round 4, with the ring-staged
    // kernels on half the chip each: ONE stream again (643 -> 634 us; knob one_wgrad_stream)
    P->single_wgrad_stream = true;
    build_plan(*P);
    Workspace ws(nullptr, 0);
    carve(*P, ws);
    P->ws_bytes = ws.used();
    P->carve_passes = 1;
    Workspace wm(nullptr, 0);
    carve(*P, wm);
    P->ws_bytes_module = wm.used();
    P->carve_passes = 3;
    return P;
}
void mm_destroy(MMPlan* P) { delete P; }
int mm_early_ranges(const MMPlan* P, long long* ranges, int cap) {
    int n = 0;
    for (const ParamInfo& pi : P->params) {
        if (pi.name.rfind("image_decoder.", 0) != 0 && pi.name.rfind("text_decoder.", 0) != 0) continue;
        if (n > 0 && ranges[2 * (n - 1)] + ranges[2 * (n - 1) + 1] == pi.offset) { ranges[2 * (n - 1) + 1] += pi.numel; continue; }
        if (n == cap) return 0;                    // (does not fit: the caller updates everything itself)
        ranges[2 * n] = pi.offset; ranges[2 * n + 1] = pi.numel; ++n;
    }
    for (int r = 0; r < n; ++r)
        if (ranges[2 * r] % 4 != 0 || ranges[2 * r + 1] % 4 != 0) return 0;
    return n;
}
int mm_D(const MMPlan* P) { return P->D; }
int mm_B(const MMPlan* P) { return P->B; }
const std::vector<ParamInfo>& mm_params(const MMPlan* P) { return P->params; }
long long mm_param_count(const MMPlan* P) { return P->nparams; }
long long mm_packed_elems(const MMPlan* P) { return P->pk.mat_elems; }
long long mm_packed_vec_elems(const MMPlan* P) { return P->pk.vec_elems > 0 ? P->pk.vec_elems : 64; }
long long mm_gpk_elems(const MMPlan* P) { return P->gk.mat_elems; }
long long mm_gpk_vec_elems(const MMPlan* P) { return P->gk.vec_elems > 0 ? P->gk.vec_elems : 64; }
int mm_ndesc(const MMPlan* P) { return (int)P->pk.d.size(); }
const PackDesc* mm_desc_host(const MMPlan* P) { return P->pk.d.data(); }
int mm_ngdesc(const MMPlan* P) { return (int)P->gk.d.size(); }
const PackDesc* mm_gdesc_host(const MMPlan* P) { return P->gk.d.data(); }
size_t mm_workspace_bytes(const MMPlan* P) { return P->ws_bytes; }
size_t mm_module_workspace_bytes(const MMPlan* P) { return P->ws_bytes_module; }

int mm_bind(MMPlan* P, const MMBuffers& b) {
    MMVAE_REQUIRE(b.params && b.grads && b.bn_stats && b.bn_nbt && b.packed && b.packed_vec && b.gpk && b.gpk_vec &&
                  b.desc_dev && b.gdesc_dev, "mm_bind: null buffer");
    P->buf = b;
    P->bound = true;
    return MMVAE_OK;
}
static int use_ws(MMPlan* P, void* ws, size_t bytes, bool module = true) {
    MMVAE_TRY(check_bound(P));
    const size_t need = module ? P->ws_bytes_module : P->ws_bytes;
    MMVAE_REQUIRE(ws != nullptr && bytes >= need, "workspace too small (%zu < %zu)", bytes, need);
    P->carve_passes = module ? 1 : 3;
    Workspace w(ws, bytes);
    carve(*P, w);
    P->dec_skip_mask = 0;
    P->slab.reset(P->w.slab, P->w.slab_floats);
    // side work or a completion event a FAILED earlier call left behind must not run against this call's buffers
    P->side_pending.clear(); P->batch_reduce = false;
    (void)mmvae_take_stop_event();
    return MMVAE_OK;
}

int mm_pack_weights(MMPlan* P, hipStream_t s) {
    MMVAE_TRY(check_bound(P));
    return launch_pack(P->buf.desc_dev, P->pk.d.data(), (int)P->pk.d.size(), P->buf.params, P->buf.packed, P->buf.packed_vec, s);
}
int mm_unpack_grads(MMPlan* P, hipStream_t s) {
    MMVAE_TRY(check_bound(P));
    MMVAE_TRY(launch_wgrad_reduce(&P->slab, s));      // partial-tile slabs -> packed gradients (no-op when none are owed)
    return launch_unpack_grads(P->buf.gdesc_dev, P->gk.d.data(), (int)P->gk.d.size(), P->buf.gpk, P->buf.gpk_vec, P->buf.grads, s);
}
int mm_grad_map(MMPlan* P, int* map, hipStream_t s) {
    MMVAE_TRY(check_bound(P));
    return launch_unpack_map(P->buf.gdesc_dev, P->gk.d.data(), (int)P->gk.d.size(), P->nparams, P->gk.mat_elems, map, s);
}
static int zero_gpk(MMPlan* P, hipStream_t s) {
    return launch_fill_zero(P->buf.gpk, (size_t)P->gk.mat_elems * sizeof(float), s);
}

static int mm_step_body(MMPlan* Pp, const MMStepIO& io, int training, int do_backward, hipStream_t s);
int mm_step_fwd_bwd(MMPlan* Pp, const MMStepIO& io, int training, int do_backward, hipStream_t s) {
    const int rc = mm_step_body(Pp, io, training, do_backward, s);
    if (rc != MMVAE_OK && Pp) join_after_error(*Pp, s);      // the message of the first error stays in mmvae_last_error
    return rc;
}
static int mm_step_body(MMPlan* Pp, const MMStepIO& io, int training, int do_backward, hipStream_t s) {
    MMVAE_TRY(use_ws(Pp, io.ws, io.ws_bytes, false));
    MMPlan& P = *Pp;
    MMPlan::W& w = P.w;
    const int B = P.B, D = P.D, B3 = 3 * B;
    MMVAE_REQUIRE(io.image && io.text && io.sums, "step: image/text/sums must be given");
    // ---- one prologue launch: zero every accumulation buffer, draw eps / dropout keep flags (Philox keyed by the
    //      device step counter) unless the caller injected them
    const float* eps = io.eps;
    const uint8_t *m1 = io.enc_mask1, *m2 = io.enc_mask2, *gk = io.gru_keep;
    // Staged prologue (knob mm_stage_begin, default on): the main stream's launch zeroes the workspace accumulators, draws the random
    // numbers and refreshes only the weight copies the image encoder's forward reads; the gradient buffers (first written in the
    // backward pass) and the other copies are done on the second-modality stream, which the main chain joins in front of the
    // product of experts anyway.  26 -> 9 us of prologue on the main chain.
    const bool serial0 = mmvae_serial();
    const bool staged = mmvae_knob("mm_stage_begin", 1) != 0 && !serial0;
    StepBeginArgs sb{}, sb2{};
    sb.zero_ptr[0] = w.zero_begin; sb.zero_bytes[0] = w.zero_bytes;
    StepBeginArgs& sz = staged ? sb2 : sb;
    if (do_backward) {
        sz.zero_ptr[1] = P.buf.gpk; sz.zero_bytes[1] = (size_t)P.gk.mat_elems * sizeof(float);
        sz.zero_ptr[2] = P.buf.grads; sz.zero_bytes[2] = (size_t)(P.nparams / 4) * 16;     // tail (< 4 floats) below
        const int skipz = mmvae_knob("dbg_begin_skip_zero", 0);      // measurement aid (results are garbage): 1 gpk, 2 grads, 3 both
        if (skipz & 1) sz.zero_ptr[1] = nullptr;
        if (skipz & 2) sz.zero_ptr[2] = nullptr;
    }
    sb.p = DROP_P; sb.seed = io.seed; sb.step = io.step_ctr;
    if (training && !eps) { sb.eps = w.eps; sb.n_eps = (long long)B3 * D; eps = w.eps; }
    if (training && io.enc_dropout && !m1) { sb.mask[0] = w.m1; sb.n_mask[0] = (long long)2 * B * 400; m1 = w.m1; }
    if (training && io.enc_dropout && !m2) { sb.mask[1] = w.m2; sb.n_mask[1] = (long long)2 * B * 200; m2 = w.m2; }
    if (training && io.gru_dropout && !gk) { sb.mask[2] = w.gkeep; sb.n_mask[2] = (long long)4 * B3 * 100; gk = w.gkeep; }
    const int skipb = mmvae_knob("dbg_begin_skip", 0);              // measurement aid (results are garbage): 1 no pack, 2 no random draws, 4 no workspace zeroing
    if (skipb & 2) { sb.eps = nullptr; sb.mask[0] = sb.mask[1] = sb.mask[2] = nullptr; }
    if (skipb & 4) sb.zero_ptr[0] = nullptr;
    const bool pack_now = io.pack_first && !(skipb & 1);
    if (pack_now) {
        MMVAE_TRY(step_begin_with_pack(sb, P.buf.desc_dev, P.pk.d.data(), (int)P.pk.d.size(), P.buf.params, P.buf.packed, P.buf.packed_vec));
        if (staged) sb.pack_parts = 1u << 0;
    }
    MMVAE_TRY(ensure_streams(P));
    P.in_step = true;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        P.capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    }
    arm_fork(P);                    // the text path forks off the prologue kernel's completion
    MMVAE_TRY(launch_step_begin(sb, s));
    MMVAE_TRY(commit_fork(P, s));
    if (do_backward && P.nparams % 4 != 0)
        MMVAE_TRY(launch_fill_zero(P.buf.grads + (P.nparams / 4) * 4, (size_t)(P.nparams % 4) * sizeof(float), s));
    const int enc_drop = training && io.enc_dropout;
    // weak-supervision variants (multimnist/paired_weak.py:84-117, modal_weak.py:87-117): an absent pass contributes no
    // loss, no gradient and no BatchNorm running-statistics update; its rows are still computed (batched with the others)
    const int sk[3] = {io.pass_skip[0] != 0, io.pass_skip[1] != 0, io.pass_skip[2] != 0};
    P.dec_skip_mask = (unsigned)(sk[0] | (sk[1] << 1) | (sk[2] << 2));
    const int enc_updates = 2 - sk[0] - sk[1];
    const bool serial = mmvae_serial();      // profiling aid: one stream, no overlap
    hipStream_t T = serial ? s : P.st_text;
    // ---- encoders: image features once for passes 1 and 2 (main), text encoder once for passes 1 and 3 (side)
    if (T != s) MMVAE_TRY(fork_to(P, T));
    if (staged && pack_now) {       // stage 1: the text copies, in front of the text encoder
        StepBeginArgs st1{};
        MMVAE_TRY(step_begin_with_pack(st1, P.buf.desc_dev, P.pk.d.data(), (int)P.pk.d.size(), P.buf.params, P.buf.packed, P.buf.packed_vec));
        st1.pack_parts = 1u << 1;
        MMVAE_TRY(launch_step_begin(st1, T));
    }
    {
        TextEncArgs a = te_args(P, io.text, w.txtout, do_backward);
        MMVAE_TRY(launch_text_encoder_fwd(a, T));
    }
    if (staged && (pack_now || sb2.zero_ptr[1] || sb2.zero_ptr[2])) {      // stage 2: gradient buffers + the copies of the decoders / the backward pass
        if (pack_now) {
            MMVAE_TRY(step_begin_with_pack(sb2, P.buf.desc_dev, P.pk.d.data(), (int)P.pk.d.size(), P.buf.params, P.buf.packed, P.buf.packed_vec));
            sb2.pack_parts = 1u << 2;
        }
        MMVAE_TRY(launch_step_begin(sb2, T));
    }
    P.step_image = io.image;
    // B % 8: the fused layers have no fallback kernel ("forced" launches) and conv4's geometry is compiled for 8 images per
    // workgroup only; other batch sizes run the unfused bn_act + gather-GEMM chain
    const bool fuse = mmvae_knob("mm_fuse_bn", 1) != 0 && mmvae_knob("convres", 1) != 0 && B % 8 == 0;
    MMVAE_TRY(enc_fwd(P, io.image, 2, m1, m2, enc_drop, training, enc_updates, w.encout, s, fuse));
    MMVAE_TRY(edge(P, T, s));
    // ---- product of experts + reparametrisation + KL for the three passes
    Latent3Args la{};
    la.B = B; la.D = D; la.img_out = w.encout; la.txt_out = w.txtout; la.eps = eps;
    la.mu = io.mu ? io.mu : w.mu; la.logvar = io.logvar ? io.logvar : w.logvar;
    la.z_f32 = w.z_f32; la.z_bf = w.z_bf; la.ldz = P.ldz; la.kl_sum = w.sums + 8; la.training = training;
    arm_fork(P);
    MMVAE_TRY(launch_latent3_fwd(la, s));
    MMVAE_TRY(commit_fork(P, s));
    // ---- text decoder (forward, backward and its weight gradients) on the side stream, image decoder on main
    if (T != s) MMVAE_TRY(fork_to(P, T));
    TextDecArgs td = td_args(P, w.z_f32, 3, do_backward);
    td.keep = (training && io.gru_dropout) ? gk : nullptr; td.keep_scale = 1.f / (1.f - DROP_P);
    td.force_tokens = io.force_tokens;
    if (io.recon_text) td.words = io.recon_text;
    if (io.tokens) td.tokens_out = io.tokens;
    td.target = io.text; td.nll_sum = w.sums + 4; td.dwords = do_backward ? w.dwords : nullptr;
    for (int k = 0; k < 3; ++k) td.nll_coef[k] = sk[k] ? 0.f : io.lambda_yx[k] / (float)(B * TXT_T);
    if (const int half = mmvae_knob("dbg_text_rows_div", 0)) td.R = max(16, td.R / half);      // measurement aid (garbage results): the text decoder on a fraction of its rows
    MMVAE_TRY(launch_text_decoder_fwd(td, T));
    if (do_backward) MMVAE_TRY(txt_dec_bwd(P, td, w.dwords, w.dz_txt, T));
    // decoders on 3B rows, BatchNorm statistics per pass; last layer fused with sigmoid + BCE (+ gradient)
    ConvTLastFwdArgs last{};
    last.target = io.image; last.recon = io.recon_image; last.dlogit = do_backward ? w.dlogit : nullptr; last.loss_sum = w.sums;
    for (int k = 0; k < 3; ++k) last.coef[k] = sk[k] ? 0.f : io.lambda_xy[k] / (float)(B * NPIX);
    // the last layer (no BatchNorm after it) of a trailing pass whose image term has weight 0 and whose reconstruction is
    // not asked for feeds nothing: multimnist/train.py:164-166 multiplies that pass's BCE by lambda_xy = 0
    int last_groups = 3;
    if (!io.recon_image)
        while (last_groups > 1 && last.coef[last_groups - 1] == 0.f) --last_groups;
    // image decoder backward runs for the leading groups with a non-zero image term (a pass with lambda_xy = 0 has exactly
    // zero gradient); the fused tail needs to know it already in the forward
    int img_groups = 3;
    while (img_groups > 0 && (io.lambda_xy[img_groups - 1] == 0.f || sk[img_groups - 1])) --img_groups;
    const bool fuse_tail = P.slab.pool != nullptr;
    MMVAE_TRY(dec_fwd(P, 3, training, &last, s, last_groups, fuse_tail ? (do_backward ? std::min(img_groups, last_groups) : 0) : -1, fuse));
    if (!do_backward) {
        MMVAE_TRY(edge(P, T, s));
        P.in_step = false;
        hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums);
        return mmvae_check_launch("sum_slots");
    }

    // =============================== backward ===============================
    // knob mm_wgrad_inline: the image path's weight gradients stay on the main stream, right behind the data gradient that made
    // their operands (no fork, no join, no contention with the main chain's kernels; 2: their partial copies summed by ONE launch
    // in front of the optimizer instead of one per layer)
    P.wgrad_forked = mmvae_knob("mm_wgrad_inline", 0) == 0;
    int rc = MMVAE_OK;
    P.deferred.clear();
    P.defer_wgrad = false;
    if (img_groups > 0) rc = dec_bwd(P, w.dlogit, img_groups, w.dz_img, s, fuse_tail, fuse, 3, training);
    // ---- early optimizer part (MMStepIO::early_adam): every gradient of image_decoder.* is complete on the weight-gradient stream (its
    //      last flush forked off the first layer's data gradient, which also closed the BatchNorm parameter gradients), text_decoder.*
    //      behind ev_txtgrads on the second-modality stream.  Their Adam update runs in the gap that stream has until the encoders'
    //      weight gradients arrive, beside the encoders' backward -- the optimizer launch behind the step shrinks to the encoders' ranges.
    if (io.early_adam && rc == MMVAE_OK && img_groups > 0 && fuse && fuse_tail && P.wgrad_forked && !serial && T != s && !io.dp_split &&
        io.defer_unpack && P.ev_txtgrads && P.side_pending.empty() && P.st_wgrad2 == P.st_wgrad && mmvae_knob("mm_early_adam", 1)) {
        hipStream_t Wst = P.st_wgrad;
        AdamArgs ad{};
        ad.p = P.buf.params; ad.g = P.buf.grads; ad.m = io.ea_m; ad.v = io.ea_v; ad.n = P.nparams; ad.step = io.ea_state;
        ad.lr = io.ea_lr; ad.b1 = io.ea_b1; ad.b2 = io.ea_b2; ad.eps = io.ea_eps; ad.grad_scale = io.ea_scale;
        ad.gmap = io.ea_gmap; ad.gpk = P.buf.gpk; ad.gpk_vec = P.buf.gpk_vec; ad.g_out = P.buf.grads;
        long long rg[8];
        ad.nr = mm_early_ranges(&P, rg, 4);
        for (int r = 0; r < ad.nr; ++r) { ad.roff[r] = rg[2 * r]; ad.rlen[r] = rg[2 * r + 1]; }
        ad.no_advance = 1;
        if (ad.nr > 0) {
            if (hipStreamWaitEvent(Wst, P.ev_txtgrads, 0) != hipSuccess) { mmvae_set_error("early optimizer part: %s", hipGetErrorString(hipGetLastError())); rc = MMVAE_EHIP; }
            if (rc == MMVAE_OK) rc = launch_adam(ad, Wst);
            if (rc == MMVAE_OK) *io.ea_ran = 1;
        }
    }
    if (rc == MMVAE_OK) {                            // dz of the text decoder
        const int k = mmvae_knob("dbg_skip_edges", 0);
        if (P.ev_dz && T != s) {
            if (k != 1 && k != 3 && hipStreamWaitEvent(s, P.ev_dz, 0) != hipSuccess) {
                mmvae_set_error("stream join failed: %s", hipGetErrorString(hipGetLastError()));
                rc = MMVAE_EHIP;
            }
        } else {
            rc = edge(P, T, s);
        }
    }
    const bool dp_split = io.dp_split && !io.defer_unpack;
    if (dp_split && rc == MMVAE_OK) {
        // Data-parallel step: every gradient of image_decoder.* and text_decoder.* has been issued -- on s (which has just
        // joined T) and on the weight-gradient streams.  T is idle until the text encoder's backward: it waits for those
        // streams, scatters the early descriptors into the flat gradient buffer and marks the event the collective waits on.
        rc = edge(P, s, T);
        if (rc == MMVAE_OK && P.st_wgrad != T) rc = edge(P, P.st_wgrad, T);
        if (rc == MMVAE_OK && P.st_wgrad2 != P.st_wgrad && P.st_wgrad2 != T) rc = edge(P, P.st_wgrad2, T);
        if (rc == MMVAE_OK) rc = launch_wgrad_reduce(&P.slab, T);
        if (rc == MMVAE_OK)
            rc = launch_unpack_grads(P.buf.gdesc_dev, P.gk.d.data(), (int)P.gk.d.size(), P.buf.gpk, P.buf.gpk_vec, P.buf.grads, T, 1);
        if (rc == MMVAE_OK) {
            if (!P.ev_early) hipEventCreateWithFlags(&P.ev_early, hipEventDisableTiming);
            if (hipEventRecord(P.ev_early, T) != hipSuccess) { mmvae_set_error("early-gradient event failed"); rc = MMVAE_EHIP; }
        }
    }
    Latent3BwdArgs lb{};
    lb.f = la; lb.dz_a = w.dz_img; lb.dz_b = w.dz_txt;
    for (int k = 0; k < 3; ++k) lb.kl_coef[k] = sk[k] ? 0.f : io.kl_lambda / (float)B;
    lb.d_img_out_bf = w.d_encout; lb.d_img_bias = P.buf.grads + P.fc[2].b_off; lb.d_txt_out = w.d_txtout;
    lb.loss_slots = w.sums; lb.loss_out = io.sums;      // every loss term is final here (text-decoder NLL joined above)
    if (rc == MMVAE_OK) {
        arm_fork(P);                // the text encoder's backward and classifier.6's weight gradient fork off this kernel
        rc = launch_latent3_bwd(lb, s);
        if (rc == MMVAE_OK) rc = commit_fork(P, s);
    }
    if (rc == MMVAE_OK) rc = flush_wgrads(P, s);
    if (rc == MMVAE_OK && T != s) rc = fork_to(P, T);
    if (rc == MMVAE_OK) rc = txt_enc_bwd(P, io.text, w.d_txtout, T);
    if (rc == MMVAE_OK) rc = enc_bwd(P, w.d_encout, 2, m1, m2, enc_drop, s, fuse);
    P.wgrad_forked = false;
    P.in_step = false;
    MMVAE_TRY(rc);
    MMVAE_TRY(join_sides(P, T, s));
    MMVAE_TRY(launch_wgrad_reduce(&P.slab, s));
    if (dp_split) return launch_unpack_grads(P.buf.gdesc_dev, P.gk.d.data(), (int)P.gk.d.size(), P.buf.gpk, P.buf.gpk_vec, P.buf.grads, s, 0);
    if (!io.defer_unpack) MMVAE_TRY(mm_unpack_grads(Pp, s));
    return MMVAE_OK;
}
int mm_wait_early_grads(MMPlan* P, hipStream_t s) {
    MMVAE_REQUIRE(P && P->ev_early, "mm_wait_early_grads: no dp_split step has run on this plan");
    if (hipStreamWaitEvent(s, P->ev_early, 0) != hipSuccess) {
        mmvae_set_error("mm_wait_early_grads: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}

// ---------------------------------------------------------------- granular module entry points (drop-in modules)
// Every call gets its own workspace (saved activations live there until the matching backward).
int mm_image_encoder_fwd(MMPlan* P, void* ws, size_t wsb, const float* image, const uint8_t* m1, const uint8_t* m2,
                         int training, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMPlan::W& w = P->w;
    MMVAE_TRY(launch_fill_zero(w.zero_begin, w.zero_bytes, s));
    return enc_fwd(*P, image, 1, m1, m2, training && m1 != nullptr, training, 1, out, s);
}
int mm_image_encoder_bwd(MMPlan* P, void* ws, size_t wsb, const float* d_out, const uint8_t* m1, const uint8_t* m2, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMPlan::W& w = P->w;
    const int rows = P->B, D2 = 2 * P->D;
    MMVAE_TRY(zero_gpk(P, s));
    hipLaunchKernelGGL(cast_bf_kernel, dim3(ceil_div(rows * D2, 256)), dim3(256), 0, s, d_out, (long long)rows * D2, w.d_encout);
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(D2, 64)), dim3(64), 0, s, d_out, rows, D2, P->buf.grads + P->fc[2].b_off);
    MMVAE_TRY(mmvae_check_launch("image encoder bwd prologue"));
    MMVAE_TRY(enc_bwd(*P, w.d_encout, 1, m1, m2, m1 != nullptr, s));
    return mm_unpack_grads(P, s);
}
int mm_image_decoder_fwd(MMPlan* P, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMPlan::W& w = P->w;
    const int rows = P->B;
    MMVAE_TRY(launch_fill_zero(w.zero_begin, w.zero_bytes, s));
    hipLaunchKernelGGL(cast_z_kernel, dim3(ceil_div(rows * P->ldz, 256)), dim3(256), 0, s, z, rows, P->D, w.z_bf, P->ldz);
    MMVAE_TRY(mmvae_check_launch("cast_z"));
    ConvTLastFwdArgs last{};
    last.recon = recon;
    return dec_fwd(*P, 1, training, &last, s);
}
int mm_image_decoder_bwd(MMPlan* P, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMPlan::W& w = P->w;
    const int rows = P->B;
    MMVAE_TRY(zero_gpk(P, s));
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ceil_div(rows * NPIX, 256)), dim3(256), 0, s, d_recon, recon, (long long)rows * NPIX, w.dlogit);
    MMVAE_TRY(mmvae_check_launch("sigmoid_bwd"));
    MMVAE_TRY(dec_bwd(*P, w.dlogit, 1, dz, s));
    return mm_unpack_grads(P, s);
}
int mm_text_encoder_fwd(MMPlan* P, void* ws, size_t wsb, const long long* text, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    TextEncArgs a = te_args(*P, text, out, true);
    return launch_text_encoder_fwd(a, s);
}
int mm_text_encoder_bwd(MMPlan* P, void* ws, size_t wsb, const long long* text, const float* d_out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(zero_gpk(P, s));
    MMVAE_TRY(txt_enc_bwd(*P, text, d_out, s));
    return mm_unpack_grads(P, s);
}
int mm_text_decoder_fwd(MMPlan* P, void* ws, size_t wsb, const float* z, int training, const uint8_t* keep,
                        const long long* force_tokens, float* words, long long* tokens, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    TextDecArgs a = td_args(*P, z, 1, true);
    a.keep = training ? keep : nullptr; a.keep_scale = 1.f / (1.f - DROP_P);
    a.force_tokens = force_tokens; a.words = words; a.tokens_out = tokens;
    return launch_text_decoder_fwd(a, s);
}
int mm_text_decoder_bwd(MMPlan* P, void* ws, size_t wsb, const float* z, const uint8_t* keep, const long long* force_tokens,
                        const float* words, const long long* tokens, const float* d_words, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(zero_gpk(P, s));
    TextDecArgs a = td_args(*P, z, 1, true);
    a.keep = keep; a.keep_scale = 1.f / (1.f - DROP_P);
    a.force_tokens = force_tokens; a.words = const_cast<float*>(words); a.tokens_out = const_cast<long long*>(tokens);
    MMVAE_TRY(txt_dec_bwd(*P, a, d_words, dz, s));
    return mm_unpack_grads(P, s);
}

// ---------------------------------------------------------------- profiling aid: replay one GEMM of the step
static bool layer_gemm(MMPlan& P, const std::string& name, GemmParams& g) {
    // the GEMMs exactly as the step launches them (epilogue options included); statistics land in the (dead)
    // accumulation buffers of the last step
    MMPlan::W& w = P.w;
    const int B = P.B;
    bf16* r[4] = {w.r1, w.r2, w.r3, w.r4};
    bf16* a[4] = {w.a1, w.a2, w.a3, w.a4};
    bf16* dre[4] = {w.d1e, w.d2e, w.d3e, w.dr4};
    bf16* q[4] = {w.u, w.q1, w.q2, w.q3};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    bf16* dq[4] = {w.du, w.d1, w.d2, w.d3};
    if (name == "enc_conv1") {
        GatherPlan pl = dense_plan(B * 625, 16, 16, 32);
        g = gemm_of(P, pl, P.conv[0].pk_fwd, 1, B * 625);
        g.c.A = w.patches1; g.out_bf = w.r1; g.ldo = 32; g.out_act_bf = w.a1; g.e_act = ACT_SWISH;
        return true;
    }
    for (int l = 1; l < 4; ++l) {
        const ConvL& L = P.conv[l];
        if (name == "enc_conv" + std::to_string(l + 1)) {
            g = gemm_of(P, L.fwd, L.pk_fwd, 1, B, L.pk_fwd_f);
            g.c.A = a[l - 1];
            g.out_bf = r[l]; g.ldo = L.g.Cout; g.colstats = w.st_e[l - 1];
            return true;
        }
        if (name == "enc_conv" + std::to_string(l + 1) + "_dgrad") {
            g = gemm_of(P, L.dgrad, L.pk_dgrad, 1, B, L.pk_dgrad_f);
            g.c.A = dre[l]; g.out_bf = dre[l - 1]; g.ldo = L.g.Cin;
            g.d_r = r[l - 1]; g.d_ld = L.g.Cin; g.d_act = ACT_SWISH;
            if (l > 1) { g.d_affine = w.aff_e[l - 2]; g.d_meanrstd = w.mr_e[l - 2]; g.d_red = w.red_e[l - 2]; }
            return true;
        }
    }
    {   // classifier / upsample Linears (2 dropout variants, masks as drawn by the last step)
        const int rows = 2 * B;
        const float ms = 1.f / (1.f - DROP_P);
        if (name == "enc_fc1") {
            GatherPlan pl = plan_fwdform(2, 2, 1, 1, 256, 2, 2, 1, 0, 400, 1, rows);
            g = gemm_of(P, pl, &P.fc[0].pk_fwd, 1, rows);
            g.c.A = w.a4; g.c.a_bcast_n = B; g.bias = P.buf.params + P.fc[0].b_off; g.out_bf = w.y1; g.ldo = 400;
            g.out_act_bf = w.ay1; g.e_act = ACT_SWISH; g.e_mask = w.m1; g.e_mask_scale = ms;
            return true;
        }
        if (name == "enc_fc2") {
            GatherPlan pl = dense_plan(rows, 400, 400, 200);
            g = gemm_of(P, pl, &P.fc[1].pk_fwd, 1, rows);
            g.c.A = w.ay1; g.bias = P.buf.params + P.fc[1].b_off; g.out_bf = w.y2; g.ldo = 200;
            g.out_act_bf = w.ay2; g.e_act = ACT_SWISH; g.e_mask = w.m2; g.e_mask_scale = ms;
            return true;
        }
        if (name == "enc_fc3") {
            GatherPlan pl = dense_plan(rows, 200, 200, 2 * P.D);
            g = gemm_of(P, pl, &P.fc[2].pk_fwd, 1, rows);
            g.c.A = w.ay2; g.bias = P.buf.params + P.fc[2].b_off; g.out_f = w.tmp_f32; g.ldo = 2 * P.D;
            return true;
        }
        if (name == "enc_fc3_dgrad") {
            GatherPlan pd = dense_plan(rows, 2 * P.D, 2 * P.D, 200);
            g = gemm_of(P, pd, &P.fc[2].pk_dgrad, 1, rows);
            g.c.A = w.d_encout; g.out_bf = w.dy2; g.ldo = 200;
            g.d_r = w.y2; g.d_ld = 200; g.d_act = ACT_SWISH; g.d_mask = w.m2; g.d_mask_scale = ms;
            g.d_colsum = w.tmp_f32;
            return true;
        }
        if (name == "enc_fc2_dgrad") {
            GatherPlan pd = dense_plan(rows, 200, 200, 400);
            g = gemm_of(P, pd, &P.fc[1].pk_dgrad, 1, rows);
            g.c.A = w.dy2; g.out_bf = w.dy1; g.ldo = 400;
            g.d_r = w.y1; g.d_ld = 400; g.d_act = ACT_SWISH; g.d_mask = w.m1; g.d_mask_scale = ms;
            g.d_colsum = w.tmp_f32;
            return true;
        }
        if (name == "dec_up") {
            GatherPlan pl = dense_plan(3 * B, P.ldz, P.ldz, 1024);
            g = gemm_of(P, pl, &P.up.pk_fwd, 1, 3 * B);
            g.c.A = w.z_bf; g.out_bf = w.u; g.ldo = 1024; g.out_act_bf = w.au; g.e_act = ACT_SWISH;
            return true;
        }
        if (name == "dec_up_dgrad") {
            GatherPlan pd = dense_plan(rows, 1024, 1024, P.D);
            g = gemm_of(P, pd, &P.up.pk_dgrad, 1, rows);
            g.c.A = w.du; g.out_f = w.tmp_f32; g.ldo = P.D;
            return true;
        }
    }
    if (name == "dec_last_dgrad_gemm") {       // thin last layer's input gradient as a dense GEMM over im2col(dlogit)
        const ConvL& L = P.convT[3];
        GatherPlan pl = dense_plan(B * 625, 16, 16, 32);
        g = gemm_of(P, pl, L.pk_dgrad, 2, B * 625);
        g.c.A = w.patches4; g.out_bf = w.d3; g.ldo = 32;
        g.d_r = w.q3; g.d_ld = 32; g.d_act = ACT_SWISH; g.d_affine = w.aff_d[2]; g.d_meanrstd = w.mr_d[2]; g.d_red = w.red_d[2];
        return true;
    }
    for (int l = 0; l < 3; ++l) {
        const ConvL& L = P.convT[l];
        if (name == "dec_convT" + std::to_string(l + 1)) {
            g = gemm_of(P, L.fwd, L.pk_fwd, 3, B, L.pk_fwd_f);
            g.c.A = aq[l];
            g.out_bf = q[l + 1]; g.ldo = L.g.Cout; g.colstats = w.st_d[l];
            return true;
        }
        if (name == "dec_convT" + std::to_string(l + 1) + "_dgrad") {
            g = gemm_of(P, L.dgrad, L.pk_dgrad, 2, B, L.pk_dgrad_f);
            g.c.A = dq[l + 1]; g.out_bf = dq[l]; g.ldo = L.g.Cin;
            g.d_r = q[l]; g.d_ld = L.g.Cin; g.d_act = ACT_SWISH;
            if (l > 0) { g.d_affine = w.aff_d[l - 1]; g.d_meanrstd = w.mr_d[l - 1]; g.d_red = w.red_d[l - 1]; }
            return true;
        }
    }
    return false;
}
// the weight gradients exactly as the step launches them (2 decoder passes carry a gradient)
static bool layer_wgrad(MMPlan& P, const std::string& name, WgradParams& g) {
    MMPlan::W& w = P.w;
    const int B = P.B;
    bf16* a[4] = {w.a1, w.a2, w.a3, w.a4};
    bf16* dre[4] = {w.d1e, w.d2e, w.d3e, w.dr4};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    bf16* dq[4] = {w.du, w.d1, w.d2, w.d3};
    for (int l = 0; l < 3; ++l)
        if (name == "dec_convT" + std::to_string(l + 1) + "_wgrad") {
            const ConvL& L = P.convT[l];
            g = convT_wgrad(P, L, 2, B, aq[l], dq[l + 1]);
            return true;
        }
    for (int l = 1; l < 4; ++l)
        if (name == "enc_conv" + std::to_string(l + 1) + "_wgrad") {
            const ConvL& L = P.conv[l];
            g = wgrad_of(P, L.fwd, L.gk, 1, B);
            g.c.A = a[l - 1]; g.P = dre[l]; g.ldp = L.g.Cout;
            return true;
        }
    if (name == "dec_last_wgrad") {
        GatherPlan pl = plan_fwdform(1, 1, 25, 25, 16, 1, 1, 1, 0, 32, 2, B);
        g = wgrad_of(P, pl, P.convT[3].gk, 2, B);
        g.c.A = w.patches4; g.c.AH = 25; g.c.AW = 25; g.c.sy = g.c.sx = 1;
        g.P = w.aq3; g.ldp = 32;
        return true;
    }
    if (name == "enc_conv1_wgrad") {
        GatherPlan pl = dense_plan(B * 625, 16, 16, 32);
        g = wgrad_of(P, pl, P.conv[0].gk, 1, B * 625);
        g.c.A = w.patches1; g.P = w.d1e; g.ldp = 32;
        return true;
    }
    return false;
}
static double wgrad_flops(const WgradParams& g) {
    double f = 0;
    for (int i = 0; i < g.c.nclasses; ++i) f += 2.0 * g.c.groups * g.cls[i].rows_per_group * (double)g.c.N * g.cls[i].K;
    return f;
}
static double gemm_flops(const GemmParams& g) {
    double f = 0;
    for (int i = 0; i < g.c.nclasses; ++i) f += 2.0 * g.c.groups * g.cls[i].rows_per_group * (double)g.c.N * g.cls[i].K;
    return f;
}
int mm_bench_layer(MMPlan* P, void* ws, size_t wsb, const char* layer, int iters, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb, false));
    if (std::string(layer) == "dec_last_fused") {        // fused decoder tail as in the training step (2 of 3 passes carry a gradient)
        ConvTLastFwdArgs last{};
        last.target = (const float*)P->w.tmp_f32; last.loss_sum = P->w.sums;
        last.coef[0] = last.coef[1] = 1.f / (float)(P->B * NPIX);
        for (int i = 0; i < iters; ++i) {
            P->slab.reset(P->w.slab, P->w.slab_floats);
            MMVAE_TRY(fused_tail(*P, 3, 1, &last, 2, 2, s));
            P->slab.jobs.clear(); P->slab.ring_jobs.clear();
        }
        return MMVAE_OK;
    }
    WgradParams wg{};
    if (layer_wgrad(*P, layer, wg)) {            // kernel + its share of the slab reduction, as in the step
        for (int i = 0; i < iters; ++i) {
            P->slab.reset(P->w.slab, P->w.slab_floats);
            MMVAE_TRY(launch_wgrad(wg, s, &P->slab));
            MMVAE_TRY(launch_wgrad_reduce(&P->slab, s));
        }
        return MMVAE_OK;
    }
    GemmParams g{};
    MMVAE_REQUIRE(layer_gemm(*P, layer, g), "bench_layer: unknown layer '%s'", layer);
    for (int i = 0; i < iters; ++i) MMVAE_TRY(launch_gemm_gather(g, s));
    return MMVAE_OK;
}
double mm_layer_flops(const MMPlan* Pc, const char* layer) {
    MMPlan& P = *const_cast<MMPlan*>(Pc);
    GemmParams g{};
    WgradParams wg{};
    if (std::string(layer) == "dec_last_fused") return 3.0 * 2.0 * 2 * P.B * 625 * 16 * 32;      // forward + both gradients, 2 passes
    if (P.bound && layer_wgrad(P, layer, wg)) return wgrad_flops(wg);
    if (!P.bound || !layer_gemm(P, layer, g)) return -1.0;
    return gemm_flops(g);
}
// FLOPs of a layer the way torch's FlopCounterMode counts the reference: 2 * out_pixels * taps * Cin * Cout for a
// Conv2d, 2 * in_pixels * taps * Cin * Cout for a ConvTranspose2d, the same number again for each of the two gradients --
// padded / cropped taps are NOT subtracted by the counter, zero-padded GEMM columns of this engine are not added.
double mm_layer_algo_flops(const MMPlan* Pc, const char* layer) {
    const MMPlan& P = *Pc;
    const std::string name = layer;
    const int B = P.B;
    auto conv = [&](const ConvL& L, int images) {
        const ConvGeom& g = L.g;
        const double pix = g.transposed ? (double)g.IH * g.IW : (double)g.OH * g.OW;
        return 2.0 * pix * g.KH * g.KW * g.Cin * g.Cout * images;
    };
    for (int l = 0; l < 4; ++l) {
        const std::string e = "enc_conv" + std::to_string(l + 1), d = "dec_convT" + std::to_string(l + 1);
        if (name == e || name == e + "_dgrad" || name == e + "_wgrad") return conv(P.conv[l], B);
        if (name == d) return conv(P.convT[l], 3 * B);
        if (name == d + "_dgrad" || name == d + "_wgrad") return conv(P.convT[l], 2 * B);
    }
    if (name == "dec_last_wgrad" || name == "dec_last_dgrad_gemm") return conv(P.convT[3], 2 * B);
    return mm_layer_flops(Pc, layer);          // dense layers: no padding in the count
}
// Algorithmic bytes of one conv-shaped layer launch: every operand tensor read once and the result written once (bf16
// activations / gradients, bf16 packed weights; the weight gradient is fp32).  What the launch would move with perfect
// reuse -- the PMC traffic of the same launch (profiles/r02_traffic.json) is compared against it.  0: not a conv layer.
double mm_layer_algo_bytes(const MMPlan* Pc, const char* layer) {
    const MMPlan& P = *Pc;
    const std::string name = layer;
    const int B = P.B;
    auto conv = [&](const ConvL& L, int images, bool wgrad) {
        const ConvGeom& g = L.g;
        const double act = (double)images * ((double)g.IH * g.IW * g.Cin + (double)g.OH * g.OW * g.Cout) * 2.0;
        return act + (double)g.Cin * g.Cout * g.KH * g.KW * (wgrad ? 4.0 : 2.0);
    };
    for (int l = 0; l < 4; ++l) {
        const std::string e = "enc_conv" + std::to_string(l + 1), d = "dec_convT" + std::to_string(l + 1);
        if (name == e || name == e + "_dgrad") return conv(P.conv[l], B, false);
        if (name == e + "_wgrad") return conv(P.conv[l], B, true);
        if (name == d) return conv(P.convT[l], 3 * B, false);
        if (name == d + "_dgrad") return conv(P.convT[l], 2 * B, false);
        if (name == d + "_wgrad") return conv(P.convT[l], 2 * B, true);
    }
    return 0.0;
}
int mm_num_bn(const MMPlan*) { return 6; }
int mm_bn_info(const MMPlan* P, int i, std::string& prefix, int& C, long long& offset) {
    if (i < 0 || i >= 6) return MMVAE_EINVAL;
    const char* bnn[6] = {"image_encoder.features.3", "image_encoder.features.6", "image_encoder.features.9",
                          "image_decoder.hallucinate.1", "image_decoder.hallucinate.4", "image_decoder.hallucinate.7"};
    prefix = bnn[i]; C = P->bn[i].C; offset = P->bn[i].stat_off;
    return MMVAE_OK;
}
long long mm_bn_floats(const MMPlan* P) { return P->bn[5].stat_off + 2 * P->bn[5].C; }

// ---------------------------------------------------------------- test aid: byte offset of a named workspace buffer
long long mm_debug_offset(MMPlan* P, const char* name) {
    Workspace ws((void*)0x1000, (size_t)1 << 40);
    P->carve_passes = 3;
    carve(*P, ws);
    MMPlan::W& w = P->w;
    std::map<std::string, const void*> m = {
        {"patches1", w.patches1}, {"r1", w.r1}, {"r2", w.r2}, {"r3", w.r3}, {"r4", w.r4}, {"y1", w.y1}, {"y2", w.y2},
        {"encout", w.encout}, {"txtout", w.txtout}, {"z_bf", w.z_bf}, {"z_f32", w.z_f32}, {"u", w.u}, {"q1", w.q1}, {"q2", w.q2},
        {"q3", w.q3}, {"logits", w.logits}, {"dlogit", w.dlogit}, {"d3", w.d3}, {"d2", w.d2}, {"d1", w.d1}, {"du", w.du},
        {"dz_img", w.dz_img}, {"dz_txt", w.dz_txt}, {"d_encout", w.d_encout}, {"d_txtout", w.d_txtout}, {"dy2", w.dy2},
        {"dy1", w.dy1}, {"db4", w.db4}, {"dr4", w.dr4}, {"d3e", w.d3e}, {"d2e", w.d2e}, {"d1e", w.d1e},
        {"aff_d0", w.aff_d[0]}, {"aff_d1", w.aff_d[1]}, {"aff_d2", w.aff_d[2]}, {"st_d0", w.st_d[0]}, {"patches4", w.patches4},
        {"tmp_f32", w.tmp_f32}, {"aff_e0", w.aff_e[0]}, {"aff_e1", w.aff_e[1]}, {"aff_e2", w.aff_e[2]},
        {"eps", w.eps}, {"m1", w.m1}, {"m2", w.m2}, {"gkeep", w.gkeep},
        {"st_e0", w.st_e[0]}, {"st_e1", w.st_e[1]}, {"st_e2", w.st_e[2]}, {"st_d1", w.st_d[1]}, {"st_d2", w.st_d[2]},
        {"red_e0", w.red_e[0]}, {"red_e1", w.red_e[1]}, {"red_e2", w.red_e[2]},
        {"red_d0", w.red_d[0]}, {"red_d1", w.red_d[1]}, {"red_d2", w.red_d[2]},
        {"a1", w.a1}, {"a2", w.a2}, {"a3", w.a3}, {"aq1", w.aq1}, {"aq2", w.aq2}, {"aq3", w.aq3},
    };
    auto it = m.find(name);
    if (it == m.end()) return -1;
    return (long long)((const char*)it->second - (const char*)0x1000);
}
