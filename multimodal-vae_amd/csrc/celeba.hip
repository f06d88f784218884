// CelebA MMVAE (celeba/model.py:14-57,91-196 ; celeba/train.py:60-81,131-147) on the shared gather-GEMM / BatchNorm /
// thin-layer kernels.  3x64x64 images, 5x5x256 bottleneck, fc 6400 <-> 1024, 18 binary attributes through
// Linear -> BatchNorm1d -> Swish MLPs, sigmoid + BCE on both modalities.
//
// Passes (celeba/train.py:138-147): (image,attrs), (image), (attrs), every lambda 1, kl_lambda 1e-3.  Image-encoder
// convolutions run once for passes 1 and 2 (classifier twice: independent Dropout masks), the attribute encoder once
// for passes 1 and 3 (no dropout: identical), decoders on 3B rows with BatchNorm statistics per pass.
#include "celeba.h"
#include "plan_base.h"
#include "thin.h"
#include <cstring>

namespace {
constexpr int IMG = 64, NPIX = 3 * IMG * IMG, NA = 18, NA_LD = 24, FEAT = 6400, HID = 1024;
}

struct CelebaPlan : PlanBase {
    int ldz;
    ConvL conv[4], convT[4];
    LinL fc1, up;
    MlpLin fc2, ae[2], ad[2];
    int fc1_dgrad;                // dense [6400][1024] packing of classifier.0 for its input gradient
    BnL bn[8];
    struct W {
        char* zero_begin; size_t zero_bytes;
        float2 *st_e[3], *red_e[3], *st_d[3], *red_d[3], *st_a[2], *red_a[2];
        float* sums; float* dz_img; float* dz_att;
        float2 *aff_e[3], *mr_e[3], *aff_d[3], *mr_d[3], *aff_a[2], *mr_a[2];
        bf16 *patches1, *r1, *r2, *r3, *r4, *a1, *a2, *a3, *a4, *y1, *ay1;
        float* encout; uint8_t* m1;
        bf16 *att_bf, *r_ae, *a_ae; float* attout;
        float *eps, *mu, *logvar, *z_f32; bf16* z_bf;
        bf16 *u, *au, *q1, *q2, *q3, *aq1, *aq2, *aq3;
        float* dlogit;
        bf16 *r_ad, *a_ad; float *alogits, *arecon, *dalogit; bf16* dalogit_bf;
        bf16 *patches4, *d3, *d2, *d1, *du, *d_ad;
        bf16 *d_encout, *d_attout_bf, *d_ae;
        bf16 *dy1, *db4, *dr4, *d3e, *d2e, *d1e;
        float* tmp_f32;
        float* slab; size_t slab_floats;
    } w;
};

namespace {

BnTabs atabs(CelebaPlan& P, int i) { CelebaPlan::W& w = P.w; return BnTabs{w.st_a[i], w.red_a[i], w.aff_a[i], w.mr_a[i]}; }

void build(CelebaPlan& P) {
    const int D = P.D;
    P.ldz = round_up(D + 1, 8);
    auto lin = [&](const std::string& n, int o, int i) { add_param(P, n + ".weight", {o, i}); add_param(P, n + ".bias", {o}); };
    auto bnp = [&](const std::string& n, int c) { add_param(P, n + ".weight", {c}); add_param(P, n + ".bias", {c}); };
    add_param(P, "image_encoder.features.0.weight", {32, 3, 4, 4});
    add_param(P, "image_encoder.features.2.weight", {64, 32, 4, 4}); bnp("image_encoder.features.3", 64);
    add_param(P, "image_encoder.features.5.weight", {128, 64, 4, 4}); bnp("image_encoder.features.6", 128);
    add_param(P, "image_encoder.features.8.weight", {256, 128, 4, 4}); bnp("image_encoder.features.9", 256);
    lin("image_encoder.classifier.0", HID, FEAT);
    lin("image_encoder.classifier.3", 2 * D, HID);
    lin("image_decoder.upsample.0", FEAT, D);
    add_param(P, "image_decoder.hallucinate.0.weight", {256, 128, 4, 4}); bnp("image_decoder.hallucinate.1", 128);
    add_param(P, "image_decoder.hallucinate.3.weight", {128, 64, 4, 4}); bnp("image_decoder.hallucinate.4", 64);
    add_param(P, "image_decoder.hallucinate.6.weight", {64, 32, 4, 4}); bnp("image_decoder.hallucinate.7", 32);
    add_param(P, "image_decoder.hallucinate.9.weight", {32, 3, 4, 4});
    lin("attrs_encoder.net.0", 64, NA); bnp("attrs_encoder.net.1", 64); lin("attrs_encoder.net.3", 2 * D, 64);
    lin("attrs_decoder.net.0", 64, D); bnp("attrs_decoder.net.1", 64); lin("attrs_decoder.net.3", NA, 64);

    const char* bnn[8] = {"image_encoder.features.3", "image_encoder.features.6", "image_encoder.features.9",
                          "image_decoder.hallucinate.1", "image_decoder.hallucinate.4", "image_decoder.hallucinate.7",
                          "attrs_encoder.net.1", "attrs_decoder.net.1"};
    const int bnc[8] = {64, 128, 256, 128, 64, 32, 64, 64};
    long long so = 0;
    for (int i = 0; i < 8; ++i) {
        P.bn[i] = BnL{off(P, std::string(bnn[i]) + ".weight"), off(P, std::string(bnn[i]) + ".bias"), bnc[i], so, i};
        P.bn_names.push_back(bnn[i]); P.bn_list.push_back(P.bn[i]);
        so += 2 * bnc[i];
    }
    // image encoder (celeba/model.py:100-112)
    build_conv(P, P.conv[0], "image_encoder.features.0.weight", ConvGeom{3, 32, 4, 4, 2, 1, 64, 64, 32, 32, false}, -1, false, true, false);
    build_conv(P, P.conv[1], "image_encoder.features.2.weight", ConvGeom{32, 64, 4, 4, 2, 1, 32, 32, 16, 16, false}, 0, true, false, false);
    build_conv(P, P.conv[2], "image_encoder.features.5.weight", ConvGeom{64, 128, 4, 4, 2, 1, 16, 16, 8, 8, false}, 1, true, false, false);
    build_conv(P, P.conv[3], "image_encoder.features.8.weight", ConvGeom{128, 256, 4, 4, 1, 0, 8, 8, 5, 5, false}, 2, true, false, false);
    // image decoder (celeba/model.py:140-151)
    build_conv(P, P.convT[0], "image_decoder.hallucinate.0.weight", ConvGeom{256, 128, 4, 4, 1, 0, 5, 5, 8, 8, true}, 3, true, false, false);
    build_conv(P, P.convT[1], "image_decoder.hallucinate.3.weight", ConvGeom{128, 64, 4, 4, 2, 1, 8, 8, 16, 16, true}, 4, true, false, false);
    build_conv(P, P.convT[2], "image_decoder.hallucinate.6.weight", ConvGeom{64, 32, 4, 4, 2, 1, 16, 16, 32, 32, true}, 5, true, false, false);
    build_conv(P, P.convT[3], "image_decoder.hallucinate.9.weight", ConvGeom{32, 3, 4, 4, 2, 1, 32, 32, 64, 64, true}, -1, true, false, true);
    add_frag_packs(P, P.conv[2]);      // 8x8 / 16x16 layers: direct-B image-resident kernels (convres.hip)
    add_frag_packs(P, P.convT[1]);

    {   // classifier.0 consumes the NCHW flatten c*25 + y*5 + x of the (256,5,5) map held here as NHWC [5][5][256]
        LinL& f = P.fc1;
        f.w_off = off(P, "image_encoder.classifier.0.weight"); f.b_off = off(P, "image_encoder.classifier.0.bias");
        f.N = HID; f.K = FEAT; f.pk_dgrad = -1;
        PackDesc d = pack_dense(f.w_off, HID, FEAT, npad_for(HID), FEAT, FEAT, 0);
        d.TW = 5; d.C = 256; d.s_ty = 5; d.s_tx = 1; d.s_c = 25;
        f.pk_fwd = P.pk.add(d);
        PackDesc gd = d; gd.Npad = round_up(HID, 64);
        f.gk = P.gk.add(gd);
        // input gradient as ONE dense GEMM: da[n][s*256+c] = sum_j dy[n][j] * W[j][c*25+s]
        PackDesc t = pack_dense(f.w_off, FEAT, HID, npad_for(FEAT), HID, 0, FEAT);
        t.NL = 256; t.s_nhi = 1; t.s_nlo = 25;
        P.fc1_dgrad = P.pk.add(t);
    }
    mlp_lin_init(P, P.fc2, "image_encoder.classifier.3", 2 * D, HID, -1, true);
    {   // upsample Linear(D, 6400): output columns permuted to NHWC n' = s*256 + c  <->  row c*25 + s; bias folded
        LinL& f = P.up;
        f.w_off = off(P, "image_decoder.upsample.0.weight"); f.b_off = off(P, "image_decoder.upsample.0.bias");
        f.N = FEAT; f.K = D;
        PackDesc d = pack_dense(f.w_off, FEAT, D, npad_for(FEAT), round_up(P.ldz, 64), 0, 1);
        d.NL = 256; d.s_nhi = D; d.s_nlo = 25 * D;
        d.bias_off = f.b_off; d.b_nhi = 1; d.b_nlo = 25;
        f.pk_fwd = P.pk.add(d);
        PackDesc gd = d; f.gk = P.gk.add(gd);
        PackDesc t = pack_dense(f.w_off, D, FEAT, npad_for(D), FEAT, 1, 0);
        t.TW = 25; t.C = 256; t.s_ty = 0; t.s_tx = D; t.s_c = 25 * D;
        f.pk_dgrad = P.pk.add(t);
    }
    mlp_lin_init(P, P.ae[0], "attrs_encoder.net.0", 64, NA, 6, false);
    mlp_lin_init(P, P.ae[1], "attrs_encoder.net.3", 2 * D, 64, -1, true);
    mlp_lin_init(P, P.ad[0], "attrs_decoder.net.0", 64, D, 7, true, P.ldz);     // operand is z_bf
    mlp_lin_init(P, P.ad[1], "attrs_decoder.net.3", NA, 64, -1, true);
}

void carve(CelebaPlan& P, Workspace& ws) {
    CelebaPlan::W& w = P.w;
    const size_t B = P.B, D = P.D, B3 = (size_t)P.carve_passes * B, B2 = (size_t)(P.carve_passes < 2 ? P.carve_passes : 2) * B;
    const int SS = MMVAE_STAT_SLOTS;
    const int ec[3] = {64, 128, 256}, dc[3] = {128, 64, 32};
    char* z0 = ws.take<char>(0);
    for (int i = 0; i < 3; ++i) { w.st_e[i] = ws.take<float2>(SS * ec[i]); w.red_e[i] = ws.take<float2>(SS * ec[i]); }
    for (int i = 0; i < 3; ++i) { w.st_d[i] = ws.take<float2>(3 * SS * dc[i]); w.red_d[i] = ws.take<float2>(3 * SS * dc[i]); }
    for (int i = 0; i < 2; ++i) { w.st_a[i] = ws.take<float2>(3 * SS * 64); w.red_a[i] = ws.take<float2>(3 * SS * 64); }
    w.sums = ws.take<float>(16 * MMVAE_LOSS_SLOTS);
    P.sk_cnt = ws.take<unsigned>(1024);
    w.dz_img = ws.take<float>(B3 * D); w.dz_att = ws.take<float>(B3 * D);
    char* z1 = ws.take<char>(0);
    w.zero_begin = z0; w.zero_bytes = (size_t)(z1 - z0);
    for (int i = 0; i < 3; ++i) { w.aff_e[i] = ws.take<float2>(ec[i]); w.mr_e[i] = ws.take<float2>(ec[i]); }
    for (int i = 0; i < 3; ++i) { w.aff_d[i] = ws.take<float2>(3 * dc[i]); w.mr_d[i] = ws.take<float2>(3 * dc[i]); }
    for (int i = 0; i < 2; ++i) { w.aff_a[i] = ws.take<float2>(3 * 64); w.mr_a[i] = ws.take<float2>(3 * 64); }
    w.patches1 = ws.take<bf16>(B * 1024 * 48);
    w.r1 = ws.take<bf16>(B * 1024 * 32); w.r2 = ws.take<bf16>(B * 256 * 64); w.r3 = ws.take<bf16>(B * 64 * 128); w.r4 = ws.take<bf16>(B * FEAT);
    w.a1 = ws.take<bf16>(B * 1024 * 32); w.a2 = ws.take<bf16>(B * 256 * 64); w.a3 = ws.take<bf16>(B * 64 * 128); w.a4 = ws.take<bf16>(B * FEAT);
    w.y1 = ws.take<bf16>(B2 * HID); w.ay1 = ws.take<bf16>(B2 * HID);
    w.encout = ws.take<float>(B2 * 2 * D); w.m1 = ws.take<uint8_t>(B2 * HID);
    w.att_bf = ws.take<bf16>(B * NA_LD); w.r_ae = ws.take<bf16>(B * 64); w.a_ae = ws.take<bf16>(B * 64); w.attout = ws.take<float>(B * 2 * D);
    w.eps = ws.take<float>(B3 * D); w.mu = ws.take<float>(B3 * D); w.logvar = ws.take<float>(B3 * D);
    w.z_f32 = ws.take<float>(B3 * D); w.z_bf = ws.take<bf16>(B3 * P.ldz);
    w.u = ws.take<bf16>(B3 * FEAT); w.au = ws.take<bf16>(B3 * FEAT);
    w.q1 = ws.take<bf16>(B3 * 64 * 128); w.q2 = ws.take<bf16>(B3 * 256 * 64); w.q3 = ws.take<bf16>(B3 * 1024 * 32);
    w.aq1 = ws.take<bf16>(B3 * 64 * 128); w.aq2 = ws.take<bf16>(B3 * 256 * 64); w.aq3 = ws.take<bf16>(B3 * 1024 * 32);
    w.dlogit = ws.take<float>(B3 * NPIX);
    w.r_ad = ws.take<bf16>(B3 * 64); w.a_ad = ws.take<bf16>(B3 * 64);
    w.alogits = ws.take<float>(B3 * NA); w.arecon = ws.take<float>(B3 * NA); w.dalogit = ws.take<float>(B3 * NA);
    w.dalogit_bf = ws.take<bf16>(B3 * NA_LD);
    w.patches4 = ws.take<bf16>(B3 * 1024 * 48);
    w.d3 = ws.take<bf16>(B3 * 1024 * 32); w.d2 = ws.take<bf16>(B3 * 256 * 64); w.d1 = ws.take<bf16>(B3 * 64 * 128);
    w.du = ws.take<bf16>(B3 * FEAT); w.d_ad = ws.take<bf16>(B3 * 64);
    w.d_encout = ws.take<bf16>(B2 * 2 * D); w.d_attout_bf = ws.take<bf16>(B * 2 * D); w.d_ae = ws.take<bf16>(B * 64);
    w.dy1 = ws.take<bf16>(B2 * HID); w.db4 = ws.take<bf16>(B2 * FEAT); w.dr4 = ws.take<bf16>(B * FEAT);
    w.d3e = ws.take<bf16>(B * 64 * 128); w.d2e = ws.take<bf16>(B * 256 * 64); w.d1e = ws.take<bf16>(B * 1024 * 32);
    w.tmp_f32 = ws.take<float>(B3 * NPIX);
    P.sk_floats = (size_t)256 * 128 * 128;
    P.sk_buf = ws.take<float>(P.sk_floats);
    // weight-gradient partial-tile slabs (written and read once per step, never zeroed): gemm.h WgradSlabCtx
    w.slab_floats = (size_t)(P.carve_passes >= 3 ? 48 : 16) << 20;
    w.slab = ws.take<float>(w.slab_floats);
}

// zero-padded bf16 copy of fp32 rows: out[r][0..ld) = (x[r][0..cols), 0...)
__global__ void cast_pad_kernel(const float* x, int rows, int cols, bf16* out, int ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ld) return;
    const int r = i / ld, c = i - r * ld;
    out[i] = (bf16)(c < cols ? x[(size_t)r * cols + c] : 0.f);
}
int cast_pad(const float* x, int rows, int cols, bf16* out, int ld, hipStream_t s) {
    hipLaunchKernelGGL(cast_pad_kernel, dim3(ceil_div(rows * ld, 256)), dim3(256), 0, s, x, rows, cols, out, ld);
    return mmvae_check_launch("cast_pad");
}

// ================================================================== image encoder (celeba/model.py:124-128)
// fuse (the fused step, B % 4 == 0): the layers that run on the image-resident kernels (convres.hip) stage BatchNorm + Swish /
// the BatchNorm backward of their gathered operand themselves (GatherTransform) and leave the staged tensor behind for the
// weight gradients (GatherTransform::out) -- no bn_act / bn_bwd_apply launch in front of them
int enc_fwd(CelebaPlan& P, const float* image, int variants, const uint8_t* m1, int dropout, int training, int bn_updates,
            float* out, hipStream_t s, bool fuse = false) {
    CelebaPlan::W& w = P.w;
    const int B = P.B;
    MMVAE_TRY(launch_im2col_small(image, B, 3, IMG, IMG, 4, 4, 2, 1, 32, 32, w.patches1, 48, s));
    {   // conv1 + Swish (no BatchNorm): raw and activated outputs
        GatherPlan pl = dense_plan(B * 1024, 48, 48, 32);
        GemmParams g = gemm_of(P, pl, P.conv[0].pk_fwd, 1, B * 1024);
        g.c.A = w.patches1; g.out_bf = w.r1; g.ldo = 32; g.out_act_bf = w.a1; g.e_act = ACT_SWISH;
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
        GatherTransform tr{};
        if (fuse && l == 2) {          // features.5 stages a2 = Swish(BatchNorm(r2)) itself
            const int prows = B * P.conv[1].g.OH * P.conv[1].g.OW;
            tr.kind = 1;
            tr.fin = bn_fin_args(P, P.bn[P.conv[1].bn], prows, 1, w.st_e[0], bn_updates, w.aff_e[0], w.mr_e[0], training);
            tr.out = a[1];
            g.c.A = r[1]; g.tr = &tr;
        }
        MMVAE_TRY(launch_gemm_gather(g, s));
        const int rows = B * L.g.OH * L.g.OW;
        if (fuse && l == 1) continue;
        MMVAE_TRY(bn_act(P, P.bn[L.bn], r[l], a[l], rows, rows, 1, w.st_e[l - 1], bn_updates, w.aff_e[l - 1], w.mr_e[l - 1], training, s));
    }
    const int rows = variants * B;
    const bool drop = training && dropout;
    {   // classifier.0 over the NHWC 5x5x256 map (shared by the variants) + Swish + Dropout
        GatherPlan pl = plan_fwdform(5, 5, 1, 1, 256, 5, 5, 1, 0, HID, 1, rows);
        GemmParams g = gemm_of(P, pl, &P.fc1.pk_fwd, 1, rows);
        g.c.A = w.a4; g.c.a_bcast_n = B;
        g.bias = P.buf.params + P.fc1.b_off; g.out_bf = w.y1; g.ldo = HID;
        g.out_act_bf = w.ay1; g.e_act = ACT_SWISH; if (drop) { g.e_mask = m1; g.e_mask_scale = 1.f / (1.f - DROP_P); }
        MMVAE_TRY(launch_gemm_gather(g, s));
    }
    return mlp_fwd(P, P.fc2, w.ay1, rows, 1, nullptr, out, nullptr, s);
}

// d_out: bf16 [variants*B][2D]; the bias gradient of classifier.3 must already be accumulated by the caller
int enc_bwd(CelebaPlan& P, const bf16* d_out, int variants, const uint8_t* m1, int dropout, hipStream_t s, bool fuse = false) {
    CelebaPlan::W& w = P.w;
    const int B = P.B, rows = variants * B;
    {   // classifier.3
        MMVAE_TRY(mlp_wgrad(P, P.fc2, d_out, w.ay1, rows, s));
        GatherPlan pd = dense_plan(rows, P.fc2.ldo, P.fc2.ldo, HID);
        GemmParams d = gemm_of(P, pd, &P.fc2.pk_dgrad, 1, rows);
        d.c.A = d_out; d.out_bf = w.dy1; d.ldo = HID;
        d.d_r = w.y1; d.d_ld = HID; d.d_act = ACT_SWISH; if (dropout) { d.d_mask = m1; d.d_mask_scale = 1.f / (1.f - DROP_P); }
        d.d_colsum = P.buf.grads + P.fc1.b_off;
        MMVAE_TRY(launch_gemm_gather(d, s));
    }
    {   // classifier.0: wgrad gathers the shared 5x5x256 map; the input gradient is one dense GEMM over NHWC columns
        GatherPlan pl = plan_fwdform(5, 5, 1, 1, 256, 5, 5, 1, 0, HID, 1, rows);
        WgradParams g = wgrad_of(P, pl, &P.fc1.gk, 1, rows);
        g.c.A = w.a4; g.c.a_bcast_n = B;
        g.P = w.dy1; g.ldp = HID;
        MMVAE_TRY(wgrad_async(P, g, s));
        GatherPlan pd = dense_plan(rows, HID, HID, FEAT);
        GemmParams d = gemm_of(P, pd, &P.fc1_dgrad, 1, rows);
        d.c.A = w.dy1; d.out_bf = w.db4; d.ldo = FEAT;
        d.d_r = w.r4; d.d_ld = FEAT; d.d_bcast_n = B; d.d_act = ACT_SWISH; d.d_cmod = 256;
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
        // features.5 / features.2 (fused): their data gradient applies the BatchNorm backward to db while staging it and writes
        // dr back IN PLACE (a workgroup owns its images: each vector is read, then overwritten, by the same thread) -- the weight
        // gradient follows it
        const bool fl = fuse && l <= 2;
        if (!fl) MMVAE_TRY(launch_bn_bwd_apply(x, s));
        WgradParams gw = wgrad_of(P, L.fwd, L.gk, 1, B);
        gw.c.A = a[l - 1]; gw.P = dr[l]; gw.ldp = L.g.Cout;
        if (!fl) MMVAE_TRY(wgrad_async(P, gw, s));
        {
            GemmParams d = gemm_of(P, L.dgrad, L.pk_dgrad, 1, B, L.pk_dgrad_f);
            d.c.A = dr[l]; d.out_bf = dr[l - 1]; d.ldo = L.g.Cin;
            d.d_r = r[l - 1]; d.d_ld = L.g.Cin; d.d_act = ACT_SWISH;
            if (l > 1) { d.d_affine = w.aff_e[l - 2]; d.d_meanrstd = w.mr_e[l - 2]; d.d_red = w.red_e[l - 2]; }
            GatherTransform tr{};
            if (fl) {
                tr.kind = 2; tr.r = r[l]; tr.red = w.red_e[l - 1]; tr.mr = w.mr_e[l - 1]; tr.gamma = P.buf.params + b.w_off;
                tr.dgamma = P.buf.grads + b.w_off; tr.dbeta = P.buf.grads + b.b_off;
                tr.inv_cnt = 1.f / (float)(B * pix); tr.groups = 1; tr.out = dr[l];
                d.tr = &tr;
            }
            MMVAE_TRY(launch_gemm_gather(d, s));
        }
        if (fl) MMVAE_TRY(wgrad_async(P, gw, s));
    }
    {   // conv1 wgrad over the im2col patches
        GatherPlan pl = dense_plan(B * 1024, 48, 48, 32);
        WgradParams g = wgrad_of(P, pl, P.conv[0].gk, 1, B * 1024);
        g.c.A = w.patches1; g.P = w.d1e; g.ldp = 32;
        MMVAE_TRY(wgrad_async(P, g, s));
    }
    return MMVAE_OK;
}

// ================================================================== image decoder (celeba/model.py:157-161)
// fused_bwd_groups >= 0 (the fused step): BatchNorm + Swish of hallucinate.7, the last ConvTranspose2d, sigmoid + BCE and -- for
// the first fused_bwd_groups passes -- its input / weight gradients in ONE kernel per image (dec_last.hip dec_last_ca_kernel);
// last_groups: passes whose last layer is computed at all
int dec_fwd(CelebaPlan& P, int groups, int training, ConvTLastFwdArgs* last, hipStream_t s, int last_groups = -1, int fused_bwd_groups = -1,
            bool fuse = false) {
    CelebaPlan::W& w = P.w;
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
        GatherTransform tr{};
        if (fuse && l >= 1) {          // hallucinate.3 / .6 stage Swish(BatchNorm(q[l])) themselves and leave aq[l] behind
            const ConvL& Lp = P.convT[l - 1];
            tr.kind = 1;
            tr.fin = bn_fin_args(P, P.bn[Lp.bn], B * Lp.g.OH * Lp.g.OW, groups, w.st_d[l - 1], 1, w.aff_d[l - 1], w.mr_d[l - 1], training);
            tr.out = aq[l];
            g.c.A = q[l]; g.tr = &tr;
        }
        MMVAE_TRY(launch_gemm_gather(g, s));
        const int rpg = B * L.g.OH * L.g.OW;
        if (l == 2 && fused_bwd_groups >= 0) continue;
        if (fuse && l <= 1) continue;
        MMVAE_TRY(bn_act(P, P.bn[L.bn], q[l + 1], aq[l + 1], groups * rpg, rpg, groups, w.st_d[l], 1, w.aff_d[l], w.mr_d[l], training, s));
    }
    if (fused_bwd_groups >= 0) {
        const ConvL& L = P.convT[2];
        const BnL& b = P.bn[L.bn];
        DecLastFusedArgs x{};
        x.r = w.q3; x.act = ACT_SWISH; x.w = P.buf.params + P.convT[3].w_off;
        x.G = last_groups > 0 ? last_groups : groups; x.B = B; x.IH = 32; x.IW = 32; x.Cin = 32; x.Cout = 3;
        x.bwd_groups = std::min(fused_bwd_groups, x.G);
        x.fin = bn_fin_args(P, b, B * L.g.OH * L.g.OW, groups, w.st_d[2], 1, w.aff_d[2], w.mr_d[2], training);
        x.target = last->target; x.logits = last->logits; x.recon = last->recon; x.dlogit = nullptr;
        for (int k = 0; k < 4; ++k) x.coef[k] = last->coef[k];
        x.loss_sum = last->loss_sum;
        if (x.bwd_groups > 0) {
            const int chunks = x.bwd_groups * B;
            x.wslab = P.slab.take((size_t)chunks * 32 * 48);
            MMVAE_REQUIRE(x.wslab != nullptr, "fused decoder tail: the weight-gradient slab pool is exhausted");
            x.db = w.d3; x.red = w.red_d[2];
            WgradSlabJob j{};
            const PackDesc& gd = P.gk.d[P.convT[3].gk[0]];
            j.dst = P.buf.gpk + gd.dst_off; j.slab = x.wslab; j.N = 32; j.K = 48; j.Kpad = gd.Kpad; j.chunks = chunks;
            j.chunk_stride = 32 * 48; j.src_ld = 48;
            j.stream = s;            // (dec_bwd moves the sum behind its first weight gradient, off the main chain)
            P.slab.jobs.push_back(j);
        }
        return launch_dec_last_ca(x, s);
    }
    ConvTLastFwdArgs x = *last;
    x.act = w.aq3; x.w = P.buf.params + P.convT[3].w_off; x.G = groups; x.B = B; x.IH = 32; x.IW = 32; x.Cin = 32; x.Cout = 3;
    return launch_convt_last_fwd(x, s);
}

// dlogit: fp32 NCHW [groups*B][3][64][64] (grad wrt the pre-sigmoid logits). Writes dz (fp32 [groups*B][D]).
int dec_bwd(CelebaPlan& P, const float* dlogit, int groups, float* dz, hipStream_t s, bool last_fused = false, bool fuse = false) {
    CelebaPlan::W& w = P.w;
    const int B = P.B, rows = groups * B;
    bf16* q[4] = {w.u, w.q1, w.q2, w.q3};
    bf16* aq[4] = {w.au, w.aq1, w.aq2, w.aq3};
    bf16* dq[4] = {w.du, w.d1, w.d2, w.d3};
    if (last_fused && P.wgrad_forked && !mmvae_serial()) {
        // d3, the BatchNorm-backward sums and the per-image weight-gradient partials of the last layer came out of the fused tail
        // (dec_fwd); the sum of the partials goes behind the first weight gradient below, on its side stream (wgrad_async)
        hipStream_t wst = (P.wgrad_rr & 1) ? P.st_wgrad2 : P.st_wgrad;
        for (WgradSlabJob& j : P.slab.jobs) if (j.stream == s && j.src_ld == 48) j.stream = wst;
    }
    if (!last_fused) {   // last transposed conv (32 -> 3): both gradients go through the im2col patches of dlogit (K = 16 taps x 3):
        // input gradient = dense GEMM patches x W with d-Swish + BatchNorm-backward sums in the epilogue
        const ConvL& L = P.convT[3];
        MMVAE_TRY(launch_im2col_small(dlogit, rows, 3, IMG, IMG, 4, 4, 2, 1, 32, 32, w.patches4, 48, s));
        {
            GatherPlan pd = dense_plan(B * 1024, 48, 48, 32);
            GemmParams d = gemm_of(P, pd, L.pk_dgrad, groups, B * 1024);
            d.c.A = w.patches4; d.out_bf = w.d3; d.ldo = 32;
            d.d_r = w.q3; d.d_ld = 32; d.d_act = ACT_SWISH; d.d_affine = w.aff_d[2]; d.d_meanrstd = w.mr_d[2]; d.d_red = w.red_d[2];
            MMVAE_TRY(launch_gemm_gather(d, s));
        }
        GatherPlan pl = plan_fwdform(1, 1, 32, 32, 48, 1, 1, 1, 0, 32, groups, B);   // rows (n, iy, ix), dense K=48
        WgradParams g = wgrad_of(P, pl, L.gk, groups, B);
        g.c.A = w.patches4; g.c.AH = 32; g.c.AW = 32; g.c.sy = g.c.sx = 1;
        g.P = w.aq3; g.ldp = 32;
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
        const bool fl = fuse && l >= 1;       // hallucinate.6 / .3: BatchNorm backward inside the data gradient's staging, dr in place (enc_bwd)
        if (!fl) MMVAE_TRY(launch_bn_bwd_apply(x, s));
        WgradParams gw = convT_wgrad(P, L, groups, B, aq[l], dq[l + 1]);
        if (!fl) MMVAE_TRY(wgrad_async(P, gw, s));
        {
            GemmParams d = gemm_of(P, L.dgrad, L.pk_dgrad, groups, B, L.pk_dgrad_f);
            d.c.A = dq[l + 1]; d.out_bf = dq[l]; d.ldo = L.g.Cin;
            d.d_r = q[l]; d.d_ld = L.g.Cin; d.d_act = ACT_SWISH;
            if (l > 0) { d.d_affine = w.aff_d[l - 1]; d.d_meanrstd = w.mr_d[l - 1]; d.d_red = w.red_d[l - 1]; }
            GatherTransform tr{};
            if (fl) {
                tr.kind = 2; tr.r = q[l + 1]; tr.red = w.red_d[l]; tr.mr = w.mr_d[l]; tr.gamma = P.buf.params + b.w_off;
                tr.dgamma = P.buf.grads + b.w_off; tr.dbeta = P.buf.grads + b.b_off;
                tr.inv_cnt = 1.f / (float)(B * pix); tr.groups = groups; tr.out = dq[l + 1];
                d.tr = &tr;
            }
            MMVAE_TRY(launch_gemm_gather(d, s));
        }
        if (fl) MMVAE_TRY(wgrad_async(P, gw, s));
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

// ================================================================== attribute MLPs (celeba/model.py:164-196)
int att_enc_fwd(CelebaPlan& P, const float* attrs, int training, int updates, float* out, hipStream_t s) {
    CelebaPlan::W& w = P.w;
    const int B = P.B;
    MMVAE_TRY(cast_pad(attrs, B, NA, w.att_bf, NA_LD, s));
    MMVAE_TRY(mlp_fwd(P, P.ae[0], w.att_bf, B, 1, w.r_ae, nullptr, training ? w.st_a[0] : nullptr, s));
    MMVAE_TRY(bn1d_act(P, P.bn[6], atabs(P, 0), w.r_ae, w.a_ae, B, 1, 64, updates, training, ACT_SWISH, s));
    return mlp_fwd(P, P.ae[1], w.a_ae, B, 1, nullptr, out, nullptr, s);
}
// d_out_bf: bf16 [B][2D]; the bias gradient of net.3 must already be accumulated by the caller
int att_enc_bwd(CelebaPlan& P, const bf16* d_out_bf, hipStream_t s) {
    CelebaPlan::W& w = P.w;
    const int B = P.B;
    BnTabs t = atabs(P, 0);
    MMVAE_TRY(mlp_wgrad(P, P.ae[1], d_out_bf, w.a_ae, B, s));
    MMVAE_TRY(mlp_dgrad(P, P.ae[1], d_out_bf, B, 1, w.d_ae, nullptr, 64, w.r_ae, &t, ACT_SWISH, s));
    MMVAE_TRY(bn1d_bwd(P, P.bn[6], t, w.d_ae, w.r_ae, B, 1, 64, s));
    return mlp_wgrad(P, P.ae[0], w.d_ae, w.att_bf, B, s);
}
int att_dec_fwd(CelebaPlan& P, int groups, int training, float* logits, hipStream_t s) {
    CelebaPlan::W& w = P.w;
    const int rows = groups * P.B;
    MMVAE_TRY(mlp_fwd(P, P.ad[0], w.z_bf, rows, groups, w.r_ad, nullptr, training ? w.st_a[1] : nullptr, s));
    MMVAE_TRY(bn1d_act(P, P.bn[7], atabs(P, 1), w.r_ad, w.a_ad, rows, groups, 64, 1, training, ACT_SWISH, s));
    return mlp_fwd(P, P.ad[1], w.a_ad, rows, 1, nullptr, logits, nullptr, s);
}
// dalogit: fp32 [groups*B][18]
int att_dec_bwd(CelebaPlan& P, const float* dalogit, int groups, float* dz, hipStream_t s) {
    CelebaPlan::W& w = P.w;
    const int rows = groups * P.B;
    BnTabs t = atabs(P, 1);
    MMVAE_TRY(cast_pad(dalogit, rows, NA, w.dalogit_bf, NA_LD, s));
    MMVAE_TRY(launch_colsum_f32(dalogit, rows, NA, P.buf.grads + P.ad[1].b_off, s));
    MMVAE_TRY(mlp_wgrad(P, P.ad[1], w.dalogit_bf, w.a_ad, rows, s));
    MMVAE_TRY(mlp_dgrad(P, P.ad[1], w.dalogit_bf, rows, groups, w.d_ad, nullptr, 64, w.r_ad, &t, ACT_SWISH, s));
    MMVAE_TRY(bn1d_bwd(P, P.bn[7], t, w.d_ad, w.r_ad, rows, groups, 64, s));
    MMVAE_TRY(mlp_wgrad(P, P.ad[0], w.d_ad, w.z_bf, rows, s));
    return mlp_dgrad(P, P.ad[0], w.d_ad, rows, 1, nullptr, dz, P.D, nullptr, nullptr, 0, s);
}

int use_ws(CelebaPlan* P, void* ws, size_t bytes, bool module = true) {
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
    return MMVAE_OK;
}
int unpack(CelebaPlan& P, hipStream_t s) {
    MMVAE_TRY(launch_wgrad_reduce(&P.slab, s));      // slab copies nobody summed yet (every side stream has joined s)
    return launch_unpack_grads(P.buf.gdesc_dev, P.gk.d.data(), (int)P.gk.d.size(), P.buf.gpk, P.buf.gpk_vec, P.buf.grads, s);
}
int zero_gpk(CelebaPlan& P, hipStream_t s) { return launch_fill_zero(P.buf.gpk, (size_t)P.gk.mat_elems * sizeof(float), s); }

}  // namespace

CelebaPlan* celeba_create(int D, int B) {
    if (D < 4 || D > 124 || D % 4 != 0 || B < 1) { mmvae_set_error("celeba_create: need n_latents in 4..124, a multiple of 4, and batch >= 1"); return nullptr; }
    CelebaPlan* P = new CelebaPlan();
    P->D = D; P->B = B;
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
void celeba_destroy(CelebaPlan* P) { delete P; }
PlanBase* celeba_base(CelebaPlan* P) { return P; }

static int celeba_step_body(CelebaPlan* Pp, const CelebaStepIO& io, int training, int do_backward, hipStream_t s);
int celeba_step(CelebaPlan* Pp, const CelebaStepIO& io, int training, int do_backward, hipStream_t s) {
    const int rc = celeba_step_body(Pp, io, training, do_backward, s);
    if (rc != MMVAE_OK && Pp) join_after_error(*Pp, s);
    return rc;
}
static int celeba_step_body(CelebaPlan* Pp, const CelebaStepIO& io, int training, int do_backward, hipStream_t s) {
    MMVAE_TRY(use_ws(Pp, io.ws, io.ws_bytes, false));
    CelebaPlan& P = *Pp;
    CelebaPlan::W& w = P.w;
    const int B = P.B, D = P.D, B3 = 3 * B;
    MMVAE_REQUIRE(io.image && io.attrs && io.sums, "celeba step: image/attrs/sums must be given");
    const float* eps = io.eps;
    const uint8_t* m1 = io.enc_mask;
    StepBeginArgs sb{};
    sb.zero_ptr[0] = w.zero_begin; sb.zero_bytes[0] = w.zero_bytes;
    if (do_backward) {
        sb.zero_ptr[1] = P.buf.gpk; sb.zero_bytes[1] = (size_t)P.gk.mat_elems * sizeof(float);
        sb.zero_ptr[2] = P.buf.grads; sb.zero_bytes[2] = (size_t)(P.nparams / 4) * 16;
    }
    sb.p = DROP_P; sb.seed = io.seed; sb.step = io.step_ctr;
    if (training && !eps) { sb.eps = w.eps; sb.n_eps = (long long)B3 * D; eps = w.eps; }
    if (training && io.enc_dropout && !m1) { sb.mask[0] = w.m1; sb.n_mask[0] = (long long)2 * B * HID; m1 = w.m1; }
    MMVAE_TRY(launch_step_begin(sb, s));
    if (do_backward && P.nparams % 4 != 0)
        MMVAE_TRY(launch_fill_zero(P.buf.grads + (P.nparams / 4) * 4, (size_t)(P.nparams % 4) * sizeof(float), s));
    const int enc_drop = training && io.enc_dropout;
    const int sk[3] = {io.pass_skip[0] != 0, io.pass_skip[1] != 0, io.pass_skip[2] != 0};
    P.dec_skip_mask = (unsigned)(sk[0] | (sk[1] << 1) | (sk[2] << 2));
    MMVAE_TRY(ensure_streams(P));
    const bool serial = mmvae_serial();
    hipStream_t T = serial ? s : P.st_text;
    // ---- encoders: attribute MLP on the side stream, image encoder on main
    MMVAE_TRY(edge(P, s, T));
    P.no_splitk = !serial;
    MMVAE_TRY(att_enc_fwd(P, io.attrs, training, 2 - sk[0] - sk[2], w.attout, T));
    P.no_splitk = false;
    // BatchNorm passes folded into the staging of the image-resident conv kernels (convres.hip; their tiles hold 4 / 2 images)
    const bool fuse = mmvae_knob("ca_fuse_bn", 1) && mmvae_knob("convres", 1) && mmvae_knob("convres_celeba", 1) && B % 4 == 0;
    MMVAE_TRY(enc_fwd(P, io.image, 2, m1, enc_drop, training, 2 - sk[0] - sk[1], w.encout, s, fuse));
    MMVAE_TRY(edge(P, T, s));
    Latent3Args la{};
    la.B = B; la.D = D; la.img_out = w.encout; la.txt_out = w.attout; la.eps = eps;
    la.mu = io.mu ? io.mu : w.mu; la.logvar = io.logvar ? io.logvar : w.logvar;
    la.z_f32 = w.z_f32; la.z_bf = w.z_bf; la.ldz = P.ldz; la.kl_sum = w.sums + 8; la.training = training;
    MMVAE_TRY(launch_latent3_fwd(la, s));
    // ---- attribute decoder (+ BCE, + its backward) on the side stream, image decoder on main
    MMVAE_TRY(edge(P, s, T));
    P.no_splitk = !serial;
    MMVAE_TRY(att_dec_fwd(P, 3, training, w.alogits, T));
    {
        BceArgs bc{};
        bc.logits = w.alogits; bc.ldl = NA; bc.target = io.attrs; bc.G = 3; bc.B = B; bc.C = NA; bc.H = 1; bc.W = 1;
        bc.recon = io.recon_attrs ? io.recon_attrs : w.arecon; bc.dlogit = do_backward ? w.dalogit : nullptr; bc.loss_sum = w.sums + 4;
        // celeba/train.py:69-73: mean over the batch per attribute, averaged over the 18 attributes
        for (int k = 0; k < 3; ++k) bc.coef[k] = sk[k] ? 0.f : io.lambda_y[k] / (float)(B * NA);
        MMVAE_TRY(launch_sigmoid_bce(bc, T));
    }
    if (do_backward) {
        P.wgrad_forked = false;      // the side stream runs its own weight gradients in order
        MMVAE_TRY(att_dec_bwd(P, w.dalogit, 3, w.dz_att, T));
    }
    P.no_splitk = false;
    ConvTLastFwdArgs last{};
    last.target = io.image; last.recon = io.recon_image; last.dlogit = do_backward ? w.dlogit : nullptr; last.loss_sum = w.sums;
    for (int k = 0; k < 3; ++k) last.coef[k] = sk[k] ? 0.f : io.lambda_x[k] / (float)(B * NPIX);
    int img_groups = 3;         // passes whose image term has a gradient (a pass with lambda_x = 0 has exactly none)
    while (img_groups > 0 && (io.lambda_x[img_groups - 1] == 0.f || sk[img_groups - 1])) --img_groups;
    int last_groups = 3;        // passes whose reconstruction is looked at by anyone
    if (!io.recon_image)
        while (last_groups > 1 && last.coef[last_groups - 1] == 0.f) --last_groups;
    const bool fuse_tail = P.slab.pool != nullptr && mmvae_knob("dec_last_ca", 1) != 0;
    MMVAE_TRY(dec_fwd(P, 3, training, &last, s, last_groups, fuse_tail ? (do_backward ? std::min(img_groups, last_groups) : 0) : -1, fuse));
    if (!do_backward) {
        MMVAE_TRY(edge(P, T, s));
        hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums);
        return mmvae_check_launch("sum_slots");
    }
    // =============================== backward ===============================
    P.wgrad_forked = true;
    int rc = MMVAE_OK;
    if (img_groups > 0) rc = dec_bwd(P, w.dlogit, img_groups, w.dz_img, s, fuse_tail, fuse);
    if (rc == MMVAE_OK) rc = edge(P, T, s);          // dz of the attribute decoder
    Latent3BwdArgs lb{};
    lb.f = la; lb.dz_a = w.dz_img; lb.dz_b = w.dz_att;
    for (int k = 0; k < 3; ++k) lb.kl_coef[k] = sk[k] ? 0.f : io.kl_lambda / (float)B;
    lb.d_img_out_bf = w.d_encout; lb.d_img_bias = P.buf.grads + P.fc2.b_off;
    lb.d_txt_out = nullptr; lb.d_txt_out_bf = w.d_attout_bf; lb.d_txt_bias = P.buf.grads + P.ae[1].b_off;
    if (rc == MMVAE_OK) rc = launch_latent3_bwd(lb, s);
    if (rc == MMVAE_OK) rc = edge(P, s, T);
    P.wgrad_forked = false; P.no_splitk = !serial;
    if (rc == MMVAE_OK) rc = att_enc_bwd(P, w.d_attout_bf, T);
    P.wgrad_forked = true; P.no_splitk = false;
    if (rc == MMVAE_OK) rc = enc_bwd(P, w.d_encout, 2, m1, enc_drop, s, fuse);
    P.wgrad_forked = false;
    MMVAE_TRY(rc);
    MMVAE_TRY(join_sides(P, T, s));
    hipLaunchKernelGGL(sum_slots_kernel, dim3(1), dim3(64), 0, s, w.sums, io.sums);
    MMVAE_TRY(mmvae_check_launch("sum_slots"));
    return io.defer_unpack ? MMVAE_OK : unpack(P, s);
}

// ---------------------------------------------------------------- granular module entry points (drop-in modules)
int celeba_image_encoder_fwd(CelebaPlan* P, void* ws, size_t wsb, const float* image, const uint8_t* mask, int training, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(launch_fill_zero(P->w.zero_begin, P->w.zero_bytes, s));
    return enc_fwd(*P, image, 1, mask, training && mask != nullptr, training, 1, out, s);
}
int celeba_image_encoder_bwd(CelebaPlan* P, void* ws, size_t wsb, const float* d_out, const uint8_t* mask, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CelebaPlan::W& w = P->w;
    const int rows = P->B, D2 = 2 * P->D;
    MMVAE_TRY(zero_gpk(*P, s));
    MMVAE_TRY(launch_cast_bf16(d_out, (long long)rows * D2, w.d_encout, s));
    MMVAE_TRY(launch_colsum_f32(d_out, rows, D2, P->buf.grads + P->fc2.b_off, s));
    MMVAE_TRY(enc_bwd(*P, w.d_encout, 1, mask, mask != nullptr, s));
    return unpack(*P, s);
}
int celeba_image_decoder_fwd(CelebaPlan* P, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CelebaPlan::W& w = P->w;
    const int rows = P->B;
    MMVAE_TRY(launch_fill_zero(w.zero_begin, w.zero_bytes, s));
    hipLaunchKernelGGL(cast_z_kernel, dim3(ceil_div(rows * P->ldz, 256)), dim3(256), 0, s, z, rows, P->D, w.z_bf, P->ldz);
    MMVAE_TRY(mmvae_check_launch("cast_z"));
    ConvTLastFwdArgs last{};
    last.recon = recon;
    return dec_fwd(*P, 1, training, &last, s);
}
int celeba_image_decoder_bwd(CelebaPlan* P, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CelebaPlan::W& w = P->w;
    const long long n = (long long)P->B * NPIX;
    MMVAE_TRY(zero_gpk(*P, s));
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, d_recon, recon, n, w.tmp_f32);
    MMVAE_TRY(mmvae_check_launch("sigmoid_bwd"));
    MMVAE_TRY(dec_bwd(*P, w.tmp_f32, 1, dz, s));
    return unpack(*P, s);
}
int celeba_attrs_encoder_fwd(CelebaPlan* P, void* ws, size_t wsb, const float* attrs, int training, float* out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    MMVAE_TRY(launch_fill_zero(P->w.zero_begin, P->w.zero_bytes, s));
    return att_enc_fwd(*P, attrs, training, 1, out, s);
}
int celeba_attrs_encoder_bwd(CelebaPlan* P, void* ws, size_t wsb, const float* d_out, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CelebaPlan::W& w = P->w;
    const int rows = P->B, D2 = 2 * P->D;
    MMVAE_TRY(zero_gpk(*P, s));
    MMVAE_TRY(launch_cast_bf16(d_out, (long long)rows * D2, w.d_attout_bf, s));
    MMVAE_TRY(launch_colsum_f32(d_out, rows, D2, P->buf.grads + P->ae[1].b_off, s));
    MMVAE_TRY(att_enc_bwd(*P, w.d_attout_bf, s));
    return unpack(*P, s);
}
int celeba_attrs_decoder_fwd(CelebaPlan* P, void* ws, size_t wsb, const float* z, int training, float* recon, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CelebaPlan::W& w = P->w;
    const int rows = P->B;
    MMVAE_TRY(launch_fill_zero(w.zero_begin, w.zero_bytes, s));
    hipLaunchKernelGGL(cast_z_kernel, dim3(ceil_div(rows * P->ldz, 256)), dim3(256), 0, s, z, rows, P->D, w.z_bf, P->ldz);
    MMVAE_TRY(mmvae_check_launch("cast_z"));
    MMVAE_TRY(att_dec_fwd(*P, 1, training, w.alogits, s));
    const long long n = (long long)rows * NA;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, w.alogits, n, recon);
    return mmvae_check_launch("sigmoid");
}
int celeba_attrs_decoder_bwd(CelebaPlan* P, void* ws, size_t wsb, const float* d_recon, const float* recon, float* dz, hipStream_t s) {
    MMVAE_TRY(use_ws(P, ws, wsb));
    CelebaPlan::W& w = P->w;
    const long long n = (long long)P->B * NA;
    MMVAE_TRY(zero_gpk(*P, s));
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, d_recon, recon, n, w.dalogit);
    MMVAE_TRY(mmvae_check_launch("sigmoid_bwd"));
    MMVAE_TRY(att_dec_bwd(*P, w.dalogit, 1, dz, s));
    return unpack(*P, s);
}
