// COCO MMVAE (coco/model.py:22-90,147-312 ; coco/train.py:66-84,146-165).
//
// Image half on the shared bf16 gather-GEMM / BatchNorm / thin-layer kernels: 3x32x32 images, four k4 s2 p1 convolutions
// (64/128/256/512 channels), classifier 2048-1024-256-2D with two Dropouts, decoder Linear(D,2048) + four transposed
// convolutions, sigmoid + BCE.  Caption half (coco_text.hip) in fp32: biGRU encoder over 102 GloVe vectors, 2-layer GRU
// decoder with vector feedback, MSE loss.
//
// Passes (coco/train.py:146-165): (image,text), (image), (text) with lambda_xy = (1,1,0), lambda_yx = (1,1,1).  Image
// features run once for passes 1 and 2 (classifier twice: independent Dropout masks), the caption encoder once for
// passes 1 and 3 (no dropout: identical), decoders on 3B rows with BatchNorm statistics per pass.  The caption half runs
// on the side stream next to the image half.
#include "coco_plan.h"
#include "thin.h"
#include <cstring>

namespace {
constexpr int IMG = 32, NPIX = 3 * IMG * IMG, FEAT = 2048, HID1 = 1024, HID2 = 256;
}

namespace {

void build(CocoPlan& P) {
    const int D = P.D;
    P.ldz = round_up(D + 1, 8);
    auto lin = [&](const std::string& n, int o, int i) { add_param(P, n + ".weight", {o, i}); add_param(P, n + ".bias", {o}); };
    auto bnp = [&](const std::string& n, int c) { add_param(P, n + ".weight", {c}); add_param(P, n + ".bias", {c}); };
    auto gru = [&](const std::string& n, const char* sfx, int in) {
        add_param(P, n + ".weight_ih_" + sfx, {COCO_G, in}); add_param(P, n + ".weight_hh_" + sfx, {COCO_G, COCO_H});
        add_param(P, n + ".bias_ih_" + sfx, {COCO_G}); add_param(P, n + ".bias_hh_" + sfx, {COCO_G});
    };
    add_param(P, "image_encoder.features.0.weight", {64, 3, 4, 4});
    add_param(P, "image_encoder.features.2.weight", {128, 64, 4, 4}); bnp("image_encoder.features.3", 128);
    add_param(P, "image_encoder.features.5.weight", {256, 128, 4, 4}); bnp("image_encoder.features.6", 256);
    add_param(P, "image_encoder.features.8.weight", {512, 256, 4, 4}); bnp("image_encoder.features.9", 512);
    lin("image_encoder.classifier.0", HID1, FEAT);
    lin("image_encoder.classifier.3", HID2, HID1);
    lin("image_encoder.classifier.6", 2 * D, HID2);
    lin("image_decoder.upsample.0", FEAT, D);
    add_param(P, "image_decoder.hallucinate.0.weight", {512, 256, 4, 4}); bnp("image_decoder.hallucinate.1", 256);
    add_param(P, "image_decoder.hallucinate.3.weight", {256, 128, 4, 4}); bnp("image_decoder.hallucinate.4", 128);
    add_param(P, "image_decoder.hallucinate.6.weight", {128, 64, 4, 4}); bnp("image_decoder.hallucinate.7", 64);
    add_param(P, "image_decoder.hallucinate.9.weight", {64, 3, 4, 4});
    gru("text_encoder.gru", "l0", COCO_E); gru("text_encoder.gru", "l0_reverse", COCO_E);
    lin("text_encoder.h2p", 2 * D, COCO_H);
    lin("text_decoder.z2h", COCO_H, D);
    gru("text_decoder.gru", "l0", COCO_E + D); gru("text_decoder.gru", "l1", COCO_H);
    lin("text_decoder.h2o", COCO_E, COCO_H + D);

    const char* bnn[6] = {"image_encoder.features.3", "image_encoder.features.6", "image_encoder.features.9",
                          "image_decoder.hallucinate.1", "image_decoder.hallucinate.4", "image_decoder.hallucinate.7"};
    const int bnc[6] = {128, 256, 512, 256, 128, 64};
    long long so = 0;
    for (int i = 0; i < 6; ++i) {
        P.bn[i] = BnL{off(P, std::string(bnn[i]) + ".weight"), off(P, std::string(bnn[i]) + ".bias"), bnc[i], so, i};
        P.bn_names.push_back(bnn[i]); P.bn_list.push_back(P.bn[i]);
        so += 2 * bnc[i];
    }
    // image encoder (coco/model.py:157-169)
    build_conv(P, P.conv[0], "image_encoder.features.0.weight", ConvGeom{3, 64, 4, 4, 2, 1, 32, 32, 16, 16, false}, -1, false, true, false);
    build_conv(P, P.conv[1], "image_encoder.features.2.weight", ConvGeom{64, 128, 4, 4, 2, 1, 16, 16, 8, 8, false}, 0, true, false, false);
    build_conv(P, P.conv[2], "image_encoder.features.5.weight", ConvGeom{128, 256, 4, 4, 2, 1, 8, 8, 4, 4, false}, 1, true, false, false);
    build_conv(P, P.conv[3], "image_encoder.features.8.weight", ConvGeom{256, 512, 4, 4, 2, 1, 4, 4, 2, 2, false}, 2, true, false, false);
    // image decoder (coco/model.py:197-208)
    build_conv(P, P.convT[0], "image_decoder.hallucinate.0.weight", ConvGeom{512, 256, 4, 4, 2, 1, 2, 2, 4, 4, true}, 3, true, false, false);
    build_conv(P, P.convT[1], "image_decoder.hallucinate.3.weight", ConvGeom{256, 128, 4, 4, 2, 1, 4, 4, 8, 8, true}, 4, true, false, false);
    build_conv(P, P.convT[2], "image_decoder.hallucinate.6.weight", ConvGeom{128, 64, 4, 4, 2, 1, 8, 8, 16, 16, true}, 5, true, false, false);
    build_conv(P, P.convT[3], "image_decoder.hallucinate.9.weight", ConvGeom{64, 3, 4, 4, 2, 1, 16, 16, 32, 32, true}, -1, true, false, true);
    add_frag_packs(P, P.conv[1]);      // 8x8 / 16x16 layers: direct-B image-resident kernels (convres.hip)
    add_frag_packs(P, P.convT[2]);
    {   // classifier.0 consumes the NCHW flatten c*4 + y*2 + x of the (512,2,2) map held here as NHWC [2][2][512]
        LinL& f = P.fc[0];
        f.w_off = off(P, "image_encoder.classifier.0.weight"); f.b_off = off(P, "image_encoder.classifier.0.bias");
        f.N = HID1; f.K = FEAT; f.pk_dgrad = -1;
        PackDesc d = pack_dense(f.w_off, HID1, FEAT, npad_for(HID1), FEAT, FEAT, 0);
        d.TW = 2; d.C = 512; d.s_ty = 2; d.s_tx = 1; d.s_c = 4;
        f.pk_fwd = P.pk.add(d);
        PackDesc gd = d; gd.Npad = round_up(HID1, 64);
        f.gk = P.gk.add(gd);
        // input gradient as ONE dense GEMM: da[n][s*512+c] = sum_j dy[n][j] * W[j][c*4+s]
        PackDesc t = pack_dense(f.w_off, FEAT, HID1, npad_for(FEAT), HID1, 0, FEAT);
        t.NL = 512; t.s_nhi = 1; t.s_nlo = 4;
        P.fc1_dgrad = P.pk.add(t);
    }
    auto dense = [&](LinL& f, const std::string& n, int N, int K) {
        f.w_off = off(P, n + ".weight"); f.b_off = off(P, n + ".bias"); f.N = N; f.K = K;
        f.pk_fwd = P.pk.add(pack_dense(f.w_off, N, K, npad_for(N), round_up(round_up(K, 8), 64), K, 1));
        f.gk = P.gk.add(pack_dense(f.w_off, N, K, round_up(N, 64), round_up(round_up(K, 8), 64), K, 1));
        f.pk_dgrad = P.pk.add(pack_dense(f.w_off, K, N, npad_for(K), round_up(round_up(N, 8), 64), 1, K));
    };
    dense(P.fc[1], "image_encoder.classifier.3", HID2, HID1);
    dense(P.fc[2], "image_encoder.classifier.6", 2 * D, HID2);
    {   // upsample Linear(D, 2048): output columns permuted to NHWC n' = s*512 + c  <->  row c*4 + s; bias folded
        LinL& f = P.up;
        f.w_off = off(P, "image_decoder.upsample.0.weight"); f.b_off = off(P, "image_decoder.upsample.0.bias");
        f.N = FEAT; f.K = D;
        PackDesc d = pack_dense(f.w_off, FEAT, D, npad_for(FEAT), round_up(P.ldz, 64), 0, 1);
        d.NL = 512; d.s_nhi = D; d.s_nlo = 4 * D;
        d.bias_off = f.b_off; d.b_nhi = 1; d.b_nlo = 4;
        f.pk_fwd = P.pk.add(d);
        PackDesc gd = d; f.gk = P.gk.add(gd);
        PackDesc t = pack_dense(f.w_off, D, FEAT, npad_for(D), FEAT, 1, 0);
        t.TW = 4; t.C = 512; t.s_ty = 0; t.s_tx = D; t.s_c = 4 * D;
        f.pk_dgrad = P.pk.add(t);
    }
    P.pk_text_begin = (int)P.pk.d.size();
    coco_text_build(P);
}

void carve(CocoPlan& P, Workspace& ws) {
    CocoPlan::W& w = P.w;
    const size_t B = P.B, D = P.D, B3 = (size_t)P.carve_passes * B, B2 = (size_t)(P.carve_passes < 2 ? P.carve_passes : 2) * B, T = P.T;
    const int SS = MMVAE_STAT_SLOTS;
    const int ec[3] = {128, 256, 512}, dc[3] = {256, 128, 64};
    char* z0 = ws.take<char>(0);
    for (int i = 0; i < 3; ++i) { w.st_e[i] = ws.take<float2>(SS * ec[i]); w.red_e[i] = ws.take<float2>(SS * ec[i]); }
    for (int i = 0; i < 3; ++i) { w.st_d[i] = ws.take<float2>(3 * SS * dc[i]); w.red_d[i] = ws.take<float2>(3 * SS * dc[i]); }
    w.sums = ws.take<float>(16 * MMVAE_LOSS_SLOTS);
    P.sk_cnt = ws.take<unsigned>(1024);
    w.dz_img = ws.take<float>(B3 * D);
    w.td_dh0 = ws.take<float>(B3 * COCO_H); w.td_dh1 = ws.take<float>(B3 * COCO_H); w.zeros_h = ws.take<float>(B3 * COCO_H);
    char* z1 = ws.take<char>(0);
    w.zero_begin = z0; w.zero_bytes = (size_t)(z1 - z0);
    w.dz_txt = ws.take<float>(B3 * D);
    for (int i = 0; i < 3; ++i) { w.aff_e[i] = ws.take<float2>(ec[i]); w.mr_e[i] = ws.take<float2>(ec[i]); }
    for (int i = 0; i < 3; ++i) { w.aff_d[i] = ws.take<float2>(3 * dc[i]); w.mr_d[i] = ws.take<float2>(3 * dc[i]); }
    w.patches1 = ws.take<bf16>(B * 256 * 48);
    w.r1 = ws.take<bf16>(B * 256 * 64); w.r2 = ws.take<bf16>(B * 64 * 128); w.r3 = ws.take<bf16>(B * 16 * 256); w.r4 = ws.take<bf16>(B * FEAT);
    w.a1 = ws.take<bf16>(B * 256 * 64); w.a2 = ws.take<bf16>(B * 64 * 128); w.a3 = ws.take<bf16>(B * 16 * 256); w.a4 = ws.take<bf16>(B * FEAT);
    w.y1 = ws.take<bf16>(B2 * HID1); w.ay1 = ws.take<bf16>(B2 * HID1); w.y2 = ws.take<bf16>(B2 * HID2); w.ay2 = ws.take<bf16>(B2 * HID2);
    w.encout = ws.take<float>(B2 * 2 * D); w.m1 = ws.take<uint8_t>(B2 * HID1); w.m2 = ws.take<uint8_t>(B2 * HID2);
    w.gkeep = ws.take<uint8_t>(T * B3 * COCO_H);
    w.eps = ws.take<float>(B3 * D); w.mu = ws.take<float>(B3 * D); w.logvar = ws.take<float>(B3 * D);
    w.z_f32 = ws.take<float>(B3 * D); w.z_bf = ws.take<bf16>(B3 * P.ldz);
    w.u = ws.take<bf16>(B3 * FEAT); w.au = ws.take<bf16>(B3 * FEAT);
    w.q1 = ws.take<bf16>(B3 * 16 * 256); w.q2 = ws.take<bf16>(B3 * 64 * 128); w.q3 = ws.take<bf16>(B3 * 256 * 64);
    w.aq1 = ws.take<bf16>(B3 * 16 * 256); w.aq2 = ws.take<bf16>(B3 * 64 * 128); w.aq3 = ws.take<bf16>(B3 * 256 * 64);
    w.dlogit = ws.take<float>(B3 * NPIX);
    w.patches4 = ws.take<bf16>(B3 * 256 * 48);
    w.d3 = ws.take<bf16>(B3 * 256 * 64); w.d2 = ws.take<bf16>(B3 * 64 * 128); w.d1 = ws.take<bf16>(B3 * 16 * 256);
    w.du = ws.take<bf16>(B3 * FEAT);
    w.d_encout = ws.take<bf16>(B2 * 2 * D); w.dy2 = ws.take<bf16>(B2 * HID2); w.dy1 = ws.take<bf16>(B2 * HID1);
    w.db4 = ws.take<bf16>(B2 * FEAT); w.dr4 = ws.take<bf16>(B * FEAT);
    w.d3e = ws.take<bf16>(B * 16 * 256); w.d2e = ws.take<bf16>(B * 64 * 128); w.d1e = ws.take<bf16>(B * 256 * 64);
    w.tmp_f32 = ws.take<float>(B3 * NPIX);
    P.sk_floats = (size_t)256 * 128 * 128;
    P.sk_buf = ws.take<float>(P.sk_floats);
    // weight-gradient partial-tile slabs (written and read once per step, never zeroed): gemm.h WgradSlabCtx
    w.slab_floats = (size_t)(P.carve_passes >= 3 ? 48 : 16) << 20;
    w.slab = ws.take<float>(w.slab_floats);
    coco_text_carve(P, ws);
}

// ================================================================== image encoder (coco/model.py:182-187)
int enc_fwd(CocoPlan& P, const float* image, int variants, const uint8_t* m1, const uint8_t* m2, int dropout, int training,
            int bn_updates, float* out, hipStream_t s) {
    CocoPlan::W& w = P.w;
    const int B = P.B;
    MMVAE_TRY(launch_im2col_small(image, B, 3, IMG, IMG, 4, 4, 2, 1, 16, 16, w.patches1, 48, s));
    {   // conv1 + Swish (no BatchNorm): raw and activated outputs
        GatherPlan pl = dense_plan(B * 256, 48, 48, 64);
        GemmParams g = gemm_of(P, pl, P.conv[0].pk_fwd, 1, B * 256);
        g.c.A = w.patches1; g.out_bf = w.r1; g.ldo = 64; g.out_act_bf = w.a1; g.e_act = ACT_SWISH;
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    bf16* r[4] = {w.r1, w.r2, w.r3, w.r4};
    bf16* a[4] = {w.a1, w.a2, w.a3, w.a4};
    for (int l = 1; l < 4; ++l) {
        const ConvL& L = P.conv[l];
        GemmParams g = gemm_of(P, L.fwd, L.pk_fwd, 1, B, L.pk_fwd_f);
        g.c.A = a[l - 1];
        g.out_bf = r[l]; g.ldo = L.g.Cout;
        g.colstats = training ? w.st_e[l - 1] : nullptr;
        MMVAE_TRY(launch_gemm_gather(g, s));
        const int rows = B * L.g.OH * L.g.OW;
        MMVAE_TRY(bn_act(P, P.bn[L.bn], r[l], a[l], rows, rows, 1, w.st_e[l - 1], bn_updates, w.aff_e[l - 1], w.mr_e[l - 1], training, s));
    }
    const int rows = variants * B;
    const bool drop = training && dropout;
    const float ms = 1.f / (1.f - DROP_P);
    {   // classifier.0 over the NHWC 2x2x512 map (shared by the variants) + Swish + Dropout
        GatherPlan pl = plan_fwdform(2, 2, 1, 1, 512, 2, 2, 1, 0, HID1, 1, rows);
        GemmParams g = gemm_of(P, pl, &P.fc[0].pk_fwd, 1, rows);
        g.c.A = w.a4; g.c.a_bcast_n = B;
        g.bias = P.buf.params + P.fc[0].b_off; g.out_bf = w.y1; g.ldo = HID1;
        g.out_act_bf = w.ay1; g.e_act = ACT_SWISH; if (drop) { g.e_mask = m1; g.e_mask_scale = ms; }
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    {
        GatherPlan pl = dense_plan(rows, HID1, HID1, HID2);
        GemmParams g = gemm_of(P, pl, &P.fc[1].pk_fwd, 1, rows);
        g.c.A = w.ay1;
        g.bias = P.buf.params + P.fc[1].b_off; g.out_bf = w.y2; g.ldo = HID2;
        g.out_act_bf = w.ay2; g.e_act = ACT_SWISH; if (drop) { g.e_mask = m2; g.e_mask_scale = ms; }
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    {
        GatherPlan pl = dense_plan(rows, HID2, HID2, 2 * P.D);
        GemmParams g = gemm_of(P, pl, &P.fc[2].pk_fwd, 1, rows);
        g.c.A = w.ay2;
        g.bias = P.buf.params + P.fc[2].b_off; g.out_f = out; g.ldo = 2 * P.D;
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    return MMVAE_OK;
}

// d_out: bf16 [variants*B][2D]; the bias gradient of classifier.6 must already be accumulated by the caller
int enc_bwd(CocoPlan& P, const bf16* d_out, int variants, const uint8_t* m1, const uint8_t* m2, int dropout, hipStream_t s) {
    CocoPlan::W& w = P.w;
    const int B = P.B, rows = variants * B, D2 = 2 * P.D;
    const float ms = 1.f / (1.f - DROP_P);
    {   // classifier.6
        GatherPlan pl = dense_plan(rows, HID2, HID2, D2);
        WgradParams g = wgrad_of(P, pl, &P.fc[2].gk, 1, rows);
        g.c.A = w.ay2; g.P = d_out; g.ldp = D2;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, D2, D2, HID2);
        GemmParams d = gemm_of(P, pd, &P.fc[2].pk_dgrad, 1, rows);
        d.c.A = d_out; d.out_bf = w.dy2; d.ldo = HID2;
        d.d_r = w.y2; d.d_ld = HID2; d.d_act = ACT_SWISH; if (dropout) { d.d_mask = m2; d.d_mask_scale = ms; }
        d.d_colsum = P.buf.grads + P.fc[1].b_off;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    {   // classifier.3
        GatherPlan pl = dense_plan(rows, HID1, HID1, HID2);
        WgradParams g = wgrad_of(P, pl, &P.fc[1].gk, 1, rows);
        g.c.A = w.ay1; g.P = w.dy2; g.ldp = HID2;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, HID2, HID2, HID1);
        GemmParams d = gemm_of(P, pd, &P.fc[1].pk_dgrad, 1, rows);
        d.c.A = w.dy2; d.out_bf = w.dy1; d.ldo = HID1;
        d.d_r = w.y1; d.d_ld = HID1; d.d_act = ACT_SWISH; if (dropout) { d.d_mask = m1; d.d_mask_scale = ms; }
        d.d_colsum = P.buf.grads + P.fc[0].b_off;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    {   // classifier.0: wgrad gathers the shared 2x2x512 map; the input gradient is one dense GEMM over NHWC columns
        GatherPlan pl = plan_fwdform(2, 2, 1, 1, 512, 2, 2, 1, 0, HID1, 1, rows);
        WgradParams g = wgrad_of(P, pl, &P.fc[0].gk, 1, rows);
        g.c.A = w.a4; g.c.a_bcast_n = B;
        g.P = w.dy1; g.ldp = HID1;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, HID1, HID1, FEAT);
        GemmParams d = gemm_of(P, pd, &P.fc1_dgrad, 1, rows);
        d.c.A = w.dy1; d.out_bf = w.db4; d.ldo = FEAT;
        d.d_r = w.r4; d.d_ld = FEAT; d.d_bcast_n = B; d.d_act = ACT_SWISH; d.d_cmod = 512;
        d.d_affine = w.aff_e[2]; d.d_meanrstd = w.mr_e[2]; d.d_red = w.red_e[2];
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    bf16* r[4] = {w.r1, w.r2, w.r3, w.r4};
    bf16* a[4] = {w.a1, w.a2, w.a3, w.a4};
    bf16* dr[4] = {w.d1e, w.d2e, w.d3e, w.dr4};
    for (int l = 3; l >= 1; --l) {
        const ConvL& L = P.conv[l];
        const BnL& b = P.bn[L.bn];
        const int pix = L.g.OH * L.g.OW;
        BnBwdApplyArgs x{};
        x.db = (l == 3) ? w.db4 : dr[l];
        x.db2 = (l == 3 && variants == 2) ? w.db4 + (size_t)B * FEAT : nullptr;
        x.r = r[l]; x.dr = dr[l]; x.rows = B * pix; x.C = L.g.Cout; x.ld = L.g.Cout; x.rows_per_group = B * pix; x.G = 1;
        x.red = w.red_e[l - 1]; x.meanrstd = w.mr_e[l - 1]; x.gamma = P.buf.params + b.w_off;
        x.dgamma = P.buf.grads + b.w_off; x.dbeta = P.buf.grads + b.b_off;
        MMVAE_TRY(launch_bn_bwd_apply(x, s));
        {
            WgradParams g = wgrad_of(P, L.fwd, L.gk, 1, B);
            g.c.A = a[l - 1]; g.P = dr[l]; g.ldp = L.g.Cout;
            MMVAE_TRY(wgrad_async(P, g, s));
        }
        {
            GemmParams d = gemm_of(P, L.dgrad, L.pk_dgrad, 1, B, L.pk_dgrad_f);
            d.c.A = dr[l]; d.out_bf = dr[l - 1]; d.ldo = L.g.Cin;
            d.d_r = r[l - 1]; d.d_ld = L.g.Cin; d.d_act = ACT_SWISH;
            if (l > 1) { d.d_affine = w.aff_e[l - 2]; d.d_meanrstd = w.mr_e[l - 2]; d.d_red = w.red_e[l - 2]; }
            MMVAE_TRY(launch_gemm_gather(d, s));
        }
    }
    {   // conv1 wgrad over the im2col patches
        GatherPlan pl = dense_plan(B * 256, 48, 48, 64);
        WgradParams g = wgrad_of(P, pl, P.conv[0].gk, 1, B * 256);
        g.c.A = w.patches1; g.P = w.d1e; g.ldp = 64;
        MMVAE_TRY(wgrad_async(P, g, s));
    }
    return MMVAE_OK;
}

// ================================================================== image decoder (coco/model.py:211-216)
int dec_fwd(CocoPlan& P, int groups, int training, ConvTLastFwdArgs* last, hipStream_t s) {
    CocoPlan::W& w = P.w;
    const int B = P.B, rows = groups * B;
    {
        GatherPlan pl = dense_plan(rows, P.ldz, P.ldz, FEAT);
        GemmParams g = gemm_of(P, pl, &P.up.pk_fwd, 1, rows);
        g.c.A = w.z_bf; g.out_bf = w.u; g.ldo = FEAT; g.out_act_bf = w.au; g.e_act = ACT_SWISH;
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    bf16* q[4] = {w.u, w.q1, w.q2, w.q3};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    for (int l = 0; l < 3; ++l) {
        const ConvL& L = P.convT[l];
        GemmParams g = gemm_of(P, L.fwd, L.pk_fwd, groups, B, L.pk_fwd_f);
        g.c.A = aq[l];
        g.out_bf = q[l + 1]; g.ldo = L.g.Cout;
        g.colstats = training ? w.st_d[l] : nullptr;
        MMVAE_TRY(launch_gemm_gather(g, s));
        const int rpg = B * L.g.OH * L.g.OW;
        MMVAE_TRY(bn_act(P, P.bn[L.bn], q[l + 1], aq[l + 1], groups * rpg, rpg, groups, w.st_d[l], 1, w.aff_d[l], w.mr_d[l], training, s));
    }
    ConvTLastFwdArgs x = *last;
    x.act = w.aq3; x.w = P.buf.params + P.convT[3].w_off; x.G = groups; x.B = B; x.IH = 16; x.IW = 16; x.Cin = 64; x.Cout = 3;
    return launch_convt_last_fwd(x, s);
}

// dlogit: fp32 NCHW [groups*B][3][32][32] (grad wrt the pre-sigmoid logits). Writes dz (fp32 [groups*B][D]).
int dec_bwd(CocoPlan& P, const float* dlogit, int groups, float* dz, hipStream_t s) {
    CocoPlan::W& w = P.w;
    const int B = P.B, rows = groups * B;
    bf16* q[4] = {w.u, w.q1, w.q2, w.q3};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    bf16* dq[4] = {w.du, w.d1, w.d2, w.d3};
    {   // last transposed conv (64 -> 3): both gradients go through the im2col patches of dlogit (K = 16 taps x 3)
        const ConvL& L = P.convT[3];
        MMVAE_TRY(launch_im2col_small(dlogit, rows, 3, IMG, IMG, 4, 4, 2, 1, 16, 16, w.patches4, 48, s));
        {
            GatherPlan pd = dense_plan(B * 256, 48, 48, 64);
            GemmParams d = gemm_of(P, pd, L.pk_dgrad, groups, B * 256);
            d.c.A = w.patches4; d.out_bf = w.d3; d.ldo = 64;
            d.d_r = w.q3; d.d_ld = 64; d.d_act = ACT_SWISH; d.d_affine = w.aff_d[2]; d.d_meanrstd = w.mr_d[2]; d.d_red = w.red_d[2];
            MMVAE_TRY(launch_gemm_gather(d, s));
        }
        GatherPlan pl = plan_fwdform(1, 1, 16, 16, 48, 1, 1, 1, 0, 64, groups, B);   // rows (n, iy, ix), dense K=48
        WgradParams g = wgrad_of(P, pl, L.gk, groups, B);
        g.c.A = w.patches4; g.c.AH = 16; g.c.AW = 16; g.c.sy = g.c.sx = 1;
        g.P = w.aq3; g.ldp = 64;
        MMVAE_TRY(wgrad_async(P, g, s));
    }
    for (int l = 2; l >= 0; --l) {
        const ConvL& L = P.convT[l];
        const BnL& b = P.bn[L.bn];
        const int pix = L.g.OH * L.g.OW;
        BnBwdApplyArgs x{};
        x.db = dq[l + 1]; x.r = q[l + 1]; x.dr = dq[l + 1]; x.rows = rows * pix; x.C = L.g.Cout; x.ld = L.g.Cout;
        x.rows_per_group = B * pix; x.G = groups;
        x.red = w.red_d[l]; x.meanrstd = w.mr_d[l]; x.gamma = P.buf.params + b.w_off;
        x.dgamma = P.buf.grads + b.w_off; x.dbeta = P.buf.grads + b.b_off;
        MMVAE_TRY(launch_bn_bwd_apply(x, s));
        {
            WgradParams g = convT_wgrad(P, L, groups, B, aq[l], dq[l + 1]);
            MMVAE_TRY(wgrad_async(P, g, s));
        }
        {
            GemmParams d = gemm_of(P, L.dgrad, L.pk_dgrad, groups, B, L.pk_dgrad_f);
            d.c.A = dq[l + 1]; d.out_bf = dq[l]; d.ldo = L.g.Cin;
            d.d_r = q[l]; d.d_ld = L.g.Cin; d.d_act = ACT_SWISH;
            if (l > 0) { d.d_affine = w.aff_d[l - 1]; d.d_meanrstd = w.mr_d[l - 1]; d.d_red = w.red_d[l - 1]; }
            MMVAE_TRY(launch_gemm_gather(d, s));
        }
    }
    {   // upsample Linear: weight (+ folded bias) gradient and dz
        GatherPlan pl = dense_plan(rows, P.ldz, P.ldz, FEAT);
        WgradParams g = wgrad_of(P, pl, &P.up.gk, 1, rows);
        g.c.A = w.z_bf; g.P = w.du; g.ldp = FEAT;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, FEAT, FEAT, P.D);
        GemmParams d = gemm_of(P, pd, &P.up.pk_dgrad, 1, rows);
        d.c.A = w.du; d.out_f = dz; d.ldo = P.D;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    return MMVAE_OK;
}

int use_ws(CocoPlan* P, void* ws, size_t bytes, bool module = true) {
    MMVAE_TRY(check_bound(P));
    const size_t need = module ? P->ws_bytes_module : P->ws_bytes;
    MMVAE_REQUIRE(ws != nullptr && bytes >= need, "workspace too small (%zu < %zu)", bytes, need);
    P->carve_passes = module ? 1 : 3;
    Workspace w(ws, bytes);
    carve(*P, w);
    P->wgrad_forked = false;
    P->dec_skip_mask = 0;
    P->slab.reset(P->w.slab, P->w.slab_floats);
    // side work or a completion event a FAILED earlier call left behind must not run against this call's buffers
    P->side_pending.clear(); P->batch_reduce = false;
    (void)mmvae_take_stop_event();
    P->dec_wg_pending = false;
    P->comb_fresh = false; P->dw16_fresh = false; P->dec_wg_composed = false;
    P->cl_alarm_f = P->cl_alarm_b = nullptr;
    return MMVAE_OK;
}
int unpack(CocoPlan& P, hipStream_t s) {
    MMVAE_TRY(launch_wgrad_reduce(&P.slab, s));      // slab copies nobody summed yet (every side stream has joined s)
    return launch_unpack_grads(P.buf.gdesc_dev, P.gk.d.data(), (int)P.gk.d.size(), P.buf.gpk, P.buf.gpk_vec, P.buf.grads, s);
}
int zero_gpk(CocoPlan& P, hipStream_t s) { return launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s); }
int zero_ws(CocoPlan& P, hipStream_t s) { return launch_fill_zero(P.w.zero_begin, P.w.zero_bytes, s); }

}  // namespace

CocoPlan* coco_create(int D, int B, int T) {
    if (D < 4 || D > 124 || D % 4 != 0 || B < 1 || T < 1 || T > 1024) {
        mmvae_set_error("coco_create: need n_latents in 4..124, a multiple of 4, batch >= 1 and 1 <= steps <= 1024");
        return nullptr;
    }
    CocoPlan* P = new CocoPlan();
    P->D = D; P->B = B; P->T = T;
    build(*P);
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
void coco_destroy(CocoPlan* P) { delete P; }
PlanBase* coco_base(CocoPlan* P) { return P; }
int coco_steps(const CocoPlan* P) { return P->T; }

static int coco_step_body(CocoPlan* Pp, const CocoStepIO& io, int training, int do_backward, hipStream_t s);
int coco_step(CocoPlan* Pp, const CocoStepIO& io, int training, int do_backward, hipStream_t s) {
    const int rc = coco_step_body(Pp, io, training, do_backward, s);
    if (rc != MMVAE_OK && Pp) join_after_error(*Pp, s);
    return rc;
}
static int coco_step_body(CocoPlan* Pp, const CocoStepIO& io, int training, int do_backward, hipStream_t s) {
    MMVAE_TRY(use_ws(Pp, io.ws, io.ws_bytes, false));
    CocoPlan& P = *Pp;
    CocoPlan::W& w = P.w;
    const int B = P.B, D = P.D, B3 = 3 * B, T = P.T;
    MMVAE_REQUIRE(io.image && io.text && io.sos && io.sums, "coco step: image/text/sos/sums must be given");
    const float* eps = io.eps;
    const uint8_t *m1 = io.enc_mask1, *m2 = io.enc_mask2, *gk = io.gru_keep;
    StepBeginArgs sb{};
    sb.zero_ptr[0] = w.zero_begin; sb.zero_bytes[0] = w.zero_bytes;
    if (do_backward) {
        sb.zero_ptr[1] = P.buf.gpk; sb.zero_bytes[1] = (size_t)P.gk.mat_elems * sizeof(float);
        sb.zero_ptr[2] = P.buf.grads; sb.zero_bytes[2] = (size_t)(P.nparams / 4) * 16;
    }
    sb.p = DROP_P; sb.seed = io.seed; sb.step = io.step_ctr;
    if (training && !eps) { sb.eps = w.eps; sb.n_eps = (long long)B3 * D; eps = w.eps; }
    if (training && io.enc_dropout && !m1) { sb.mask[0] = w.m1; sb.n_mask[0] = (long long)2 * B * HID1; m1 = w.m1; }
    if (training && io.enc_dropout && !m2) { sb.mask[1] = w.m2; sb.n_mask[1] = (long long)2 * B * HID2; m2 = w.m2; }
    if (training && io.gru_dropout && !gk) { sb.mask[2] = w.gkeep; sb.n_mask[2] = (long long)T * B3 * COCO_H; gk = w.gkeep; }
    if (!(training && io.gru_dropout)) gk = nullptr;
    MMVAE_TRY(launch_step_begin(sb, s));
    if (do_backward && P.nparams % 4 != 0)
        MMVAE_TRY(launch_fill_zero(P.buf.grads + (P.nparams / 4) * 4, (size_t)(P.nparams % 4) * sizeof(float), s));
    const int enc_drop = training && io.enc_dropout;
    const int sk[3] = {io.pass_skip[0] != 0, io.pass_skip[1] != 0, io.pass_skip[2] != 0};
    P.dec_skip_mask = (unsigned)(sk[0] | (sk[1] << 1) | (sk[2] << 2));
    MMVAE_TRY(ensure_streams(P));
    const bool serial = mmvae_serial();
    hipStream_t Tx = serial ? s : P.st_text;
    // ---- encoders: caption GRU on the side stream, image encoder on main
    MMVAE_TRY(edge(P, s, Tx));
    hipStream_t Sd = serial ? s : P.st_wgrad;       // idle until the backward pass: the caption decoder's packs
    const bool packing = io.pack_first && !P.no_pack;
    if (packing) {      // the caption encoder's recurrence opens the critical chain: its (few) packs first, everything else beside them
        const int nd = (int)P.pk.d.size();
        auto range = [&](int d0, int d1, hipStream_t st) {
            return launch_pack_range(P.buf.desc_dev, P.pk.d.data(), nd, d0, d1, P.buf.params, P.buf.packed, P.buf.packed_vec, st);
        };
        MMVAE_TRY(range(P.pk_text_begin, P.pk_textdec_begin, Tx));
        if (Sd != s) MMVAE_TRY(edge(P, s, Sd));
        MMVAE_TRY(range(P.pk_textdec_begin, nd, Sd));
        MMVAE_TRY(range(0, P.pk_text_begin, s));
    }
    MMVAE_TRY(coco_text_enc_fwd(P, io.text, do_backward, w.txtout, Tx, true, serial ? nullptr : &s));
    MMVAE_TRY(enc_fwd(P, io.image, 2, m1, m2, enc_drop, training, 2 - sk[0] - sk[1], w.encout, s));
    if (coco_text_dec_composed(P, B3)) MMVAE_TRY(coco_text_dec_prepare(P, io.sos, s));     // (main stream: idle here until the caption encoder is through)
    MMVAE_TRY(edge(P, Tx, s));
    Latent3Args la{};
    la.B = B; la.D = D; la.img_out = w.encout; la.txt_out = w.txtout; la.eps = eps;
    la.mu = io.mu ? io.mu : w.mu; la.logvar = io.logvar ? io.logvar : w.logvar;
    la.z_f32 = w.z_f32; la.z_bf = w.z_bf; la.ldz = P.ldz; la.kl_sum = w.sums + 8; la.training = training;
    MMVAE_TRY(launch_latent3_fwd(la, s));
    // ---- caption decoder (+ MSE, + its backward) on the side stream, image decoder on main
    MMVAE_TRY(edge(P, s, Tx));
    if (packing && Sd != Tx) MMVAE_TRY(edge(P, Sd, Tx));
    float* sentence = io.recon_text ? io.recon_text : w.td_recon;
    {
        CocoMseFuse mf{};
        for (int k = 0; k < 3; ++k) mf.coef[k] = sk[k] ? 0.f : io.lambda_yx[k] / ((float)B * (float)T * (float)COCO_E);
        mf.target = io.text; mf.loss_sum = w.sums; mf.dw = do_backward ? w.td_dw : nullptr; mf.dw16 = P.text_bf16 ? w.tb_dw16 : nullptr;
        MMVAE_TRY(coco_text_dec_fwd(P, w.z_f32, 3, io.sos, gk, do_backward, sentence, Tx, true, &mf));
        if (!P.mse_fused)       // (no bf16 copy of the gradient here: only the composed BPTT kernel reads one, and converts it itself then)
            MMVAE_TRY(coco_mse3(sentence, io.text, 3, (long long)B * T * COCO_E, mf.coef, w.sums, mf.dw, Tx));
        P.dw16_fresh = do_backward && P.mse_fused;
    }
    if (do_backward) MMVAE_TRY(coco_text_dec_bwd(P, w.z_f32, 3, io.sos, gk, sentence, w.td_dw, w.dz_txt, Tx, serial ? Tx : P.st_wgrad2, true));
    ConvTLastFwdArgs last{};
    last.target = io.image; last.recon = io.recon_image; last.dlogit = do_backward ? w.dlogit : nullptr; last.loss_sum = w.sums;
    for (int k = 0; k < 3; ++k) last.coef[k] = sk[k] ? 0.f : io.lambda_xy[k] / (float)(B * NPIX);
    MMVAE_TRY(dec_fwd(P, 3, training, &last, s));
    if (!do_backward) {
        MMVAE_TRY(edge(P, Tx, s));
        hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums, P.cl_alarm_f, (const unsigned*)nullptr);
        return mmvae_check_launch("sum_slots");
    }
    // =============================== backward ===============================
    P.wgrad_forked = true;
    int img_groups = 3;
    while (img_groups > 0 && (io.lambda_xy[img_groups - 1] == 0.f || sk[img_groups - 1])) --img_groups;
    int rc = MMVAE_OK;
    if (img_groups > 0) rc = dec_bwd(P, w.dlogit, img_groups, w.dz_img, s);
    if (rc == MMVAE_OK) rc = edge(P, Tx, s);          // dz of the caption decoder
    Latent3BwdArgs lb{};
    lb.f = la; lb.dz_a = w.dz_img; lb.dz_b = w.dz_txt;
    for (int k = 0; k < 3; ++k) lb.kl_coef[k] = sk[k] ? 0.f : io.kl_lambda / (float)B;
    lb.d_img_out_bf = w.d_encout; lb.d_img_bias = P.buf.grads + P.fc[2].b_off;
    lb.d_txt_out = w.d_txtout;
    if (rc == MMVAE_OK) rc = launch_latent3_bwd(lb, s);
    if (rc == MMVAE_OK) rc = edge(P, s, Tx);
    if (rc == MMVAE_OK) rc = coco_text_enc_bwd(P, io.text, w.d_txtout, Tx, serial ? Tx : P.st_wgrad2, true);
    if (rc == MMVAE_OK) rc = coco_text_dec_wgrads(P, serial ? Tx : P.st_wgrad2);   // (no-op: issued behind the encoder's BPTT launch)
    if (rc == MMVAE_OK) rc = enc_bwd(P, w.d_encout, 2, m1, m2, enc_drop, s);
    P.wgrad_forked = false;
    MMVAE_TRY(rc);
    MMVAE_TRY(edge(P, Tx, s));
    MMVAE_TRY(edge(P, P.st_wgrad, s));
    if (P.st_wgrad2 != P.st_wgrad) MMVAE_TRY(edge(P, P.st_wgrad2, s));
    if (io.defer_unpack) MMVAE_TRY(launch_wgrad_reduce(&P.slab, s));  // the packed gradients are complete; Adam gathers them
    else MMVAE_TRY(unpack(P, s));
    // (timeout words of the decoder's cluster launches of THIS step: zeroed before each of them, set only when an exchange gave up;
    //  last kernel of the step, so that the mark in the gradient is not overwritten)
    hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums, P.cl_alarm_f, P.cl_alarm_b, io.optimizer_state, P.buf.grads);
    return mmvae_check_launch("sum_slots");
}

// ---------------------------------------------------------------- granular module entry points (drop-in modules)
int coco_image_encoder_fwd(CocoPlan* P, void* ws, size_t wsb, const float* image, const uint8_t* m1, const uint8_t* m2, int training,
                           float* out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(zero_ws(*P, s));
    return enc_fwd(*P, image, 1, m1, m2, training && m1 != nullptr && m2 != nullptr, training, 1, out, s);
}
int coco_image_encoder_bwd(CocoPlan* P, void* ws, size_t wsb, const float* d_out, const uint8_t* m1, const uint8_t* m2, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CocoPlan::W& w = P->w;
    const int rows = P->B, D2 = 2 * P->D;
    MMVAE_TRY(zero_gpk(*P, s));
    MMVAE_TRY(launch_cast_bf16(d_out, (long long)rows * D2, w.d_encout, s));
    MMVAE_TRY(launch_colsum_f32(d_out, rows, D2, P->buf.grads + P->fc[2].b_off, s));
    MMVAE_TRY(enc_bwd(*P, w.d_encout, 1, m1, m2, m1 != nullptr && m2 != nullptr, s));
    return unpack(*P, s);
}
int coco_image_decoder_fwd(CocoPlan* P, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CocoPlan::W& w = P->w;
    const int rows = P->B;
    MMVAE_TRY(zero_ws(*P, s));
    hipLaunchKernelGGL(cast_z_kernel, dim3(ceil_div(rows * P->ldz, 256)), dim3(256), 0, s, z, rows, P->D, w.z_bf, P->ldz);
    MMVAE_TRY(mmvae_check_launch("cast_z"));
    ConvTLastFwdArgs last{};
    last.recon = recon;
    return dec_fwd(*P, 1, training, &last, s);
}
int coco_image_decoder_bwd(CocoPlan* P, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CocoPlan::W& w = P->w;
    const long long n = (long long)P->B * NPIX;
    MMVAE_TRY(zero_gpk(*P, s));
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, d_recon, recon, n, w.tmp_f32);
    MMVAE_TRY(mmvae_check_launch("sigmoid_bwd"));
    MMVAE_TRY(dec_bwd(*P, w.tmp_f32, 1, dz, s));
    return unpack(*P, s);
}
int coco_text_encoder_fwd(CocoPlan* P, void* ws, size_t wsb, const float* text, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(zero_ws(*P, s));
    return coco_text_enc_fwd(*P, text, 1, out, s, false);
}
int coco_text_encoder_bwd(CocoPlan* P, void* ws, size_t wsb, const float* text, const float* d_out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    return coco_text_enc_bwd(*P, text, d_out, s, s, false);
}
int coco_text_decoder_fwd(CocoPlan* P, void* ws, size_t wsb, const float* z, const float* sos, const uint8_t* keep, int training,
                          float* sentence, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(zero_ws(*P, s));
    return coco_text_dec_fwd(*P, z, 1, sos, training ? keep : nullptr, 1, sentence, s);
}
int coco_text_decoder_bwd(CocoPlan* P, void* ws, size_t wsb, const float* z, const float* sos, const uint8_t* keep, const float* sentence,
                          const float* d_sentence, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CocoPlan::W& w = P->w;
    const size_t n = (size_t)P->B * P->T * COCO_E;
    MMVAE_TRY(launch_fill_zero(w.td_dh0, (size_t)P->B * COCO_H * sizeof(float), s));
    MMVAE_TRY(launch_fill_zero(w.td_dh1, (size_t)P->B * COCO_H * sizeof(float), s));
    hipMemcpyAsync(w.td_dw, d_sentence, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    return coco_text_dec_bwd(*P, z, 1, sos, keep, sentence, w.td_dw, dz, s, s);
}
