// Private: the COCO plan object shared by coco.hip (image half, step, entry points) and coco_text.hip (caption GRUs, fp32).
#pragma once
#include "coco.h"
#include "plan_base.h"

constexpr int COCO_E = 300, COCO_H = 200, COCO_G = 3 * COCO_H;

struct CocoGru { long long wih, whh, bih, bhh; };

// bf16 persistent caption decoder (coco_text_bf16.hip): padded operand widths
constexpr int CTB_HP = 224, CTB_XP = 320, CTB_GP = 608, CTB_EP = 304;

struct CocoDecFwdArgs {
    int R, T;
    const float* hinit;      // [R][200] z2h(z)
    const float* zi0;        // [R][600] z-part of the layer-0 input projection (+ b_ih)
    const float* zo;         // [R][300] z-part of the output projection (+ bias)
    const float* sos;        // [300]
    const uint8_t* keep; float keep_scale;        // [T][R][200] inter-layer dropout keep flags or null
    const bf16 *w_ih0, *w_hh0, *w_ih1, *w_hh1, *w_ho;       // packed [608][320], [608][224] x3, [304][224]
    // cluster form (coco_dec_fwd_cl_kernel): per-gate packs [3][208][K] of the four GRU matrices, exchange granules
    // [row blocks][3 phases][16][200] (zeroed before the launch), timeout word; cluster = ranks per row block (0 / 1: off)
    const bf16 *wg_ih0, *wg_hh0, *wg_ih1, *wg_hh1;
    // composed form (coco_dec_fwd_c8_kernel, cluster == 8): per-gate packs [3][208][224] of W_comb = W_ih0[:, :300] W_ho[:, :200],
    // zi0p = zi0 + zo W_ih0[:, :300]^T ([R][600]), sosv = W_ih0[:, :300] sos ([600]); null: the three-exchange form
    const bf16* wg_comb; const float *zi0p, *sosv;
    // composed form only, optional: the reconstruction loss of the 3 passes fused into the output pass.  mse_target [B][T][300]
    // (rows of pass g = g*B ..), mse_coef[g] the loss weights, mse_loss the loss slots (slot*16 + 4 + g, += squared error),
    // mse_dw [R][T][300] / mse_dw16 [R][T][320] the loss gradient (null: loss only)
    const float* mse_target; float mse_coef[3]; int mse_B; float* mse_loss; float* mse_dw; bf16* mse_dw16;
    unsigned long long* cl_xchg; unsigned* cl_timeout; int cluster;   // (cl_xchg: raw bytes, 16-byte aligned)
    const float *bhh0, *bih1, *bhh1;
    float* sentence;         // [R][T][300]
    // saved for the backward pass (null: inference)
    float *h0_all, *h1_all;  // [T+1][R][200] (index 0 = hinit, written by the caller)
    float *sav0, *sav1;      // [T][R][4*200] (r, z, n, W_hn h + b_hn)
    bf16 *xb_all, *h0b_all, *midb_all, *h1b_all;           // bf16 GEMM operands in [t][row] layout for the batched weight gradients
};
struct CocoDecBwdArgs {
    int R, T;
    const float* dw;         // [R][T][300] loss gradient wrt the sentence
    const uint8_t* keep; float keep_scale;
    const bf16 *w_hoT, *w_ih1T, *w_hh1T, *w_hh0T, *w_ih0T;  // packed [208][320], [208][608] x3, [304][608]
    const float *h0_all, *h1_all, *sav0, *sav1;
    bf16 *dout_b;            // [T][R][304] total gradient wrt each step's output
    bf16 *dgi0_b, *dgh0_b, *dgi1_b, *dgh1_b;               // [T][R][608]
    float *dhinit;           // [R][200]
    float *dwsum;            // [R][300] time sum of the output gradient
    // cluster form (coco_dec_bwd_cl_kernel): exchange granules [row blocks][2*2*16*200 + 16*150] (zeroed before the launch)
    unsigned long long* cl_xchg; unsigned* cl_timeout; int cluster;   // (cl_xchg: raw bytes, 16-byte aligned)
    // composed form (coco_dec_bwd_c8_kernel, cluster == 8; null: three exchanges): packed [208][608] W_comb^T and the bf16 copy
    // [R][T][320] of dw (pad columns zero).  That kernel leaves dout_b to the caller (dOut[t] = dw[t] + dgi0[t+1] W_ih0x) and
    // writes only the time sum of dw into dwsum
    const bf16 *w_combT, *dw16;
    float *dzi0, *dzi1;      // composed form: [R][600] time sum of dgi0 over all steps / over t >= 1 (made by the kernel)
};
struct CocoEncFwdArgs {
    int B, T;
    const float* gi;         // [B][T][600] input projection of every step (+ b_ih); resident form: [600][T][B] WITHOUT b_ih
    const float* bih;        // resident form: b_ih, added in fp32 by the kernel
    const bf16* w_hh;        // packed [608][224]; resident form: three per-gate [208][224] matrices back to back
    int resident;            // 1: the weights stay in registers / LDS for the whole recurrence (coco_enc_fwd_res_kernel);
                             // gi is then [600][T][B] (no bias), sav [T][4][200][B], h_all [T][200][B] (batch row fastest, B % 4 == 0)
    float* h_last;           // resident form: h after the last step, [B][200]
    const float* bhh;
    float* h_all;            // [T][B][200] h after each step
    float* sav;              // [T][B][4*200] (r, z, n, W_hn h + b_hn) or null (inference)
    bf16* hb_all;            // [T][B][224] bf16 h BEFORE each step (slice 0 = zeros), column 200 = 1.0: operand of the batched weight gradient
};
struct CocoEncBwdArgs {
    int B, T;
    const float* dh_init;    // [B][200] gradient wrt h[T-1]
    const float *sav, *h_all;
    const bf16* w_hhT;       // packed [208][608]
    int resident;            // 1: coco_enc_bwd_res_kernel (sav / h_all in the layouts of the resident forward kernel)
    bf16 *dgi_b, *dgh_b;     // [T][B][608] gradients wrt the input / hidden projections
};
int launch_coco_enc_fwd(const CocoEncFwdArgs& a, hipStream_t s);
int launch_coco_enc_bwd(const CocoEncBwdArgs& a, hipStream_t s);
// dst[(t*B + b)*ld + e] = bf16(text[(b*T + t)*300 + e]); column 300 = 1.0
int launch_coco_text_tb(const float* text, int B, int T, int ld, bf16* dst, hipStream_t s);
int launch_coco_dec_fwd(const CocoDecFwdArgs& a, hipStream_t s);
// comb: [3][208][224], combT: [208][608] (fragment-major bf16), sosv: [600], wz: [600][D] = W_ih0z + W_ih0x W_hoz, bz: [600] =
// b_ih + W_ih0x b_ho; wih0 [600][300 + D] / who [300][200 + D] and their biases: the fp32 parameters
int launch_coco_comb(const float* wih0, const float* bih0, const float* who, const float* bho, int D, const float* sos, bf16* comb, bf16* combT,
                     float* sosv, float* wz, float* bz, hipStream_t s);
int launch_coco_dec_bwd(const CocoDecBwdArgs& a, hipStream_t s);
// out[r][c] (fp32, [R][cols]) = sum over t of in[(t*R + r)*ld + c]
int launch_coco_time_sum_bf16(const bf16* in, int T, int R, int ld, int cols, float* out, hipStream_t s);
// dout[(t*R + r)*304 + e] = bf16(dw[(r*T + t)*300 + e] + (t < T-1 ? fb[(t*R + r)*300 + e] : 0)): total gradient wrt each step's output
int launch_coco_dout_combine(const float* dw, const float* fb, int T, int R, bf16* dout, hipStream_t s);
// dw16[r*320 + e] = bf16(dw[r*300 + e]), pad columns zero (rows = R*T)
int launch_coco_dw16(const float* dw, long long rows, bf16* dw16, hipStream_t s);

struct CocoPlan : PlanBase {
    int ldz, T;
    ConvL conv[4], convT[4];
    LinL fc[3], up;
    int fc1_dgrad;
    BnL bn[6];
    // caption half: offsets into the flat fp32 parameter buffer (the GEMMs read the weights where they are)
    CocoGru te_f, te_r, td0, td1;
    long long te_h2p_w, te_h2p_b, td_z2h_w, td_z2h_b, td_h2o_w, td_h2o_b;
    // bf16 persistent caption decoder: packed weights (forward and transposed forms) and packed-gradient descriptors
    bool text_bf16 = true;
    const unsigned *cl_alarm_f = nullptr, *cl_alarm_b = nullptr;   // timeout words of this step's cluster launches (null: not used)
    const float* dec_wg_dw = nullptr;
    bool dw16_fresh = false, dec_wg_composed = false;             // bf16 copy of dw made by this step's decoder forward (fused MSE); dOut left to the wgrads
    int tb_ih0xT_rm = -1;                                          // row-major pack of W_ih0[:, :300]^T ([300][600]) for the batched dOut GEMM
    int pk_textdec_begin = 0;                                      // ... of the caption DECODER (the encoder's come first)
    int pk_text_begin = 0;                                         // first pack descriptor of the caption half (the table's tail)
    bool mse_fused = false;                                        // the last caption-decoder forward computed the MSE terms itself
    bool comb_fresh = false;                                       // W_comb / sosv made from the CURRENT parameters (reset by use_ws)
    bool dec_wg_pending = false; const float* dec_wg_z = nullptr; int dec_wg_groups = 0;   // deferred weight gradients of the bf16 decoder
    int tb_ih0, tb_hh0, tb_ih1, tb_hh1, tb_ho, tb_hoT, tb_ih1T, tb_hh1T, tb_hh0T, tb_ih0T, tb_e_hh, tb_e_hhT, tb_e_ihA, tb_e_hhg[3], tb_g_ih0[3], tb_g_hh0[3], tb_g_ih1[3], tb_g_hh1[3];
    int tg_ih0, tg_hh0, tg_ih1, tg_hh1, tg_ho, tg_e_ih, tg_e_hh;
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
        float* slab; size_t slab_floats;
        // caption encoder (B rows)
        float *te_gi, *te_gh, *te_h, *te_sav, *te_gi_r, *te_sav_r, *te_hb, *te_sum, *txtout;
        float *d_txtout, *te_dsum, *te_dh, *te_dgi, *te_dgh, *te_dgi_r;
        // caption decoder (3B rows)
        float *td_zi0, *td_zo, *td_gi, *td_gh, *td_h0, *td_h1, *td_mid, *td_sav0, *td_sav1, *td_recon;
        float *td_dw, *td_dgi0, *td_dgh0, *td_dgi1, *td_dgh1, *td_dmid, *td_dzi0, *td_dwsum, *td_dhinit;
        // bf16 persistent decoder: operands of the batched weight gradients, [t][row] layout
        bf16 *tb_x, *tb_h0, *tb_mid, *tb_h1, *tb_dout, *tb_dgi0, *tb_dgh0, *tb_dgi1, *tb_dgh1;
        bf16 *te_xb, *te_hb_all, *te_dgi_b, *te_dgh_b;
        float *te_giT, *te_hlast;
        bf16 *tb_comb, *tb_combT; float *td_sosv, *td_zi0p;
        bf16* tb_dw16; float* td_dzi1;
        float *td_wz, *td_bz;
        char* cl_xchg; size_t cl_bytes; char* clb_xchg; size_t clb_bytes;
    } w;
};

// coco_text.hip
void coco_text_build(CocoPlan& P);                      // parameter offsets (the parameters are added by coco.hip's build)
void coco_text_carve(CocoPlan& P, Workspace& ws);       // the non-zeroed caption buffers
// text: [B][T][300]; out: [B][2D]
// issues the caption decoder's deferred weight gradients (bf16 path) on `sw`; no-op when none are pending
int coco_text_dec_wgrads(CocoPlan& P, hipStream_t sw);
// composed decoder weights (W_comb, sosv) from the current parameters, once per step, on any stream the decoder's stream is
// ordered behind; the decoder makes them itself (on its own stream) when nobody did
int coco_text_dec_prepare(CocoPlan& P, const float* sos, hipStream_t s);
bool coco_text_dec_composed(const CocoPlan& P, int R);      // will a decoder pass over R rows run the composed kernels?
// side (or null): a stream ordered behind the step's prologue on which the reverse direction's single step runs
int coco_text_enc_fwd(CocoPlan& P, const float* text, int save, float* out, hipStream_t s, bool bf16_path, const hipStream_t* side = nullptr);
// d_out: [B][2D] (the h2p bias gradient is added here); accumulates every caption-encoder gradient into P.buf.grads
int coco_text_enc_bwd(CocoPlan& P, const float* text, const float* d_out, hipStream_t s, hipStream_t sw, bool bf16_path);
// z: [rows][D] fp32, rows = groups*B; sentence: [rows][T][300]; keep: [T][rows][200] or null
// mse (or null): target / loss weights / outputs of the reconstruction loss; when the kernel that runs can fuse it into its
// output pass (composed cluster form) it does, and P.mse_fused tells the caller not to launch coco_mse3
struct CocoMseFuse { const float* target; float coef[3]; float* loss_sum; float* dw; bf16* dw16; };
int coco_text_dec_fwd(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, int save, float* sentence, hipStream_t s,
                      bool bf16_path = false, const CocoMseFuse* mse = nullptr);
// dw: [rows][T][300] gradient wrt the sentence (consumed: the feedback gradients are accumulated into it); dz: [rows][D]
// sw: stream for the weight gradients (== s: in order; a side stream: the caller joins it before reading the gradients)
int coco_text_dec_bwd(CocoPlan& P, const float* z, int groups, const float* sos, const uint8_t* keep, const float* sentence, float* dw, float* dz, hipStream_t s, hipStream_t sw,
                      bool bf16_path = false);
// recon [G*B][T][300] vs target [B][T][300]: loss_sum[slot][4+g] += sum sq err ; dw = coef[g] * 2 (recon - target) (or null)
int coco_mse3(const float* recon, const float* target, int G, long long per_group, const float* coef, float* loss_sum, float* dw, hipStream_t s);
