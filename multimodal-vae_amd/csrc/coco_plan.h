// Private: the COCO plan object shared by coco.hip (image half, step, entry points) and coco_text.hip (caption GRUs, fp32).
#pragma once
#include "coco.h"
#include "plan_base.h"

constexpr int COCO_E = 300, COCO_H = 200, COCO_G = 3 * COCO_H;

struct CocoGru { long long wih, whh, bih, bhh; };

struct CocoPlan : PlanBase {
    int ldz, T;
    ConvL conv[4], convT[4];
    LinL fc[3], up;
    int fc1_dgrad;
    BnL bn[6];
    // caption half: offsets into the flat fp32 parameter buffer (the GEMMs read the weights where they are)
    CocoGru te_f, te_r, td0, td1;
    long long te_h2p_w, te_h2p_b, td_z2h_w, td_z2h_b, td_h2o_w, td_h2o_b;
    struct W {
        char* zero_begin; size_t zero_bytes;
        float2 *st_e[3], *red_e[3], *st_d[3], *red_d[3];
        float* sums; float* dz_img; float* dz_txt; float* td_dh0; float* td_dh1; float* zeros_h;
        float2 *aff_e[3], *mr_e[3], *aff_d[3], *mr_d[3];
        bf16 *patches1, *r1, *r2, *r3, *r4, *a1, *a2, *a3, *a4, *y1, *ay1, *y2, *ay2;
        float* encout; uint8_t *m1, *m2, *gkeep;
        float *eps, *mu, *logvar, *z_f32; bf16* z_bf;
        bf16 *u, *au, *q1, *q2, *q3, *aq1, *aq2, *aq3;
        float* dlogit;
        bf16 *patches4, *d3, *d2, *d1, *du;
        bf16 *d_encout, *dy2, *dy1, *db4, *dr4, *d3e, *d2e, *d1e;
        float* tmp_f32;
        // caption encoder (B rows)
        float *te_gi, *te_gh, *te_h, *te_sav, *te_gi_r, *te_sav_r, *te_hb, *te_sum, *txtout;
        float *d_txtout, *te_dsum, *te_dh, *te_dgi, *te_dgh, *te_dgi_r;
        // caption decoder (3B rows)
        float *td_zi0, *td_zo, *td_gi, *td_gh, *td_h0, *td_h1, *td_mid, *td_sav0, *td_sav1, *td_recon;
        float *td_dw, *td_dgi0, *td_dgh0, *td_dgi1, *td_dgh1, *td_dmid, *td_dzi0, *td_dwsum, *td_dhinit;
    } w;
};

// coco_text.hip
void coco_text_build(CocoPlan& P);                      // parameter offsets (the parameters are added by coco.hip's build)
void coco_text_carve(CocoPlan& P, Workspace& ws);       // the non-zeroed caption buffers
// text: [B][T][300]; out: [B][2D]
int coco_text_enc_fwd(CocoPlan& P, const float* text, int save, float* out, hipStream_t s);
// d_out: [B][2D] (the h2p bias gradient is added here); accumulates every caption-encoder gradient into P.buf.grads
int coco_text_enc_bwd(CocoPlan& P, const float* text, const float* d_out, hipStream_t s);
// z: [rows][D] fp32, rows = groups*B; sentence: [rows][T][300]; keep: [T][rows][200] or null
int coco_text_dec_fwd(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, int save, float* sentence, hipStream_t s);
// dw: [rows][T][300] gradient wrt the sentence (consumed: the feedback gradients are accumulated into it); dz: [rows][D]
// sw: stream for the weight gradients (== s: in order; a side stream: the caller joins it before reading the gradients)
int coco_text_dec_bwd(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, const float* sentence, float* dw, float* dz, hipStream_t s, hipStream_t sw);
// recon [G*B][T][300] vs target [B][T][300]: loss_sum[slot][4+g] += sum sq err ; dw = coef[g] * 2 (recon - target) (or null)
int coco_mse3(const float* recon, const float* target, int G, long long per_group, const float* coef, float* loss_sum, float* dw, hipStream_t s);
