// MultiMNIST text path (multimnist/model.py:219-307): persistent row-tile kernels, one launch per direction.
// Each workgroup owns 16 batch rows (rows are independent: no BatchNorm on this path) and runs the whole
// recurrence for them; every matmul is a 16-row MFMA tile against packed bf16 weights streamed from L2.
#pragma once
#include "common.h"

#define TXT_H 100          // GRU hidden size (multimnist/model.py:26,29)
#define TXT_HP 128         // hidden size padded to an MFMA K multiple
#define TXT_G3 300         // 3*H gate columns
#define TXT_G3P 304        // padded to 16
#define TXT_G3K 320        // padded to 32 (as a reduction dim)
#define TXT_T 4            // max_length (multimnist/utils.py:14)
#define TXT_V 12           // n_characters

struct GruPacked {
    const bf16* wih;  int kih;    // [304][kih]   (gate-major, K = input features padded to 32)
    const bf16* whh;              // [304][128]
    const bf16* wihT; int nih;    // [nih][320]   transposed copy for backward (nih = input features padded to 16)
    const bf16* whhT;             // [112][320]
    const float* bih; const float* bhh;   // fp32 [300]
};

struct TextEncArgs {
    int B, D;
    const long long* tokens;       // [B][4]
    const float* embed;            // [12][100] fp32
    GruPacked fwd, rev;
    const bf16* h2p;  int nh2p;    // [round16(2D)][128]
    const bf16* h2pT;              // [112][round32(2D)]
    const float* h2p_bias;         // [2D]
    float* out;                    // [B][2D]  (mu | logvar)
    // saved for backward (may be null in eval)
    float* gates_f;                // [4][5][B][100]  r,z,n,ghn,hprev
    float* gates_r;                // [5][B][100]
    bf16* x_bf;                    // [4][B][128] embedded tokens
    bf16* hprev_bf;                // [4][B][128]
    bf16* hsum_bf;                 // [B][128]
};
int launch_text_encoder_fwd(const TextEncArgs& a, hipStream_t s);

struct TextEncBwdArgs {
    TextEncArgs f;
    const float* d_out;            // [B][2D]
    bf16* d_out_bf;                // [B][round8(2D)]  (P operand of the h2p wgrad)
    bf16* dgi_f; bf16* dgh_f;      // [4][B][304]
    bf16* dgi_r;                   // [B][304]
    float* g_embed;                // [12][100] +=
    float* g_bih_f; float* g_bhh_f; float* g_bih_r; float* g_bhh_r;   // [300] +=
    float* g_h2p_bias;             // [2D] +=
};
int launch_text_encoder_bwd(const TextEncBwdArgs& a, hipStream_t s);

struct TextDecArgs {
    int R, D;                      // rows (= passes*B), latent size
    int rows_per_pass;             // B (for the per-pass NLL sums)
    const float* z;                // [R][D]
    const float* embed;            // [12][100]
    const bf16* z2h;  int kz;      // [112][kz]  kz = round32(D)
    const bf16* z2hT;              // [round16(D)][128]
    const float* z2h_bias;         // [100]
    GruPacked l0, l1;              // l0 input = 100 + D features
    const bf16* h2o;  int kx;      // [16][kx]   kx = round32(100+D)
    const bf16* h2oT;              // [round16(100+D)][32]
    const float* h2o_bias;         // [12]
    const uint8_t* keep;           // [4][R][100] inter-layer dropout keep flags, null = no dropout
    float keep_scale;
    const long long* force_tokens; // [R][4] or null (test hook: overrides the greedy feedback)
    float* words;                  // [R][4][12] log-probs
    long long* tokens_out;         // [R][4] greedy argmax, may be null
    // fused NLL (multimnist/train.py:79): target tokens shared by all passes
    const long long* target;       // [rows_per_pass][4] or null
    float* nll_sum;                // [passes] += sum of -logp[target]
    float* dwords;                 // [R][4][12] = coef[pass] * dNLL/dlogp, or null
    float nll_coef[4];
    // saved for backward (null in eval)
    float* gates;                  // [4][2][5][R][100]
    bf16* x0_bf;                   // [4][R][kx]
    bf16* h0p_bf; bf16* mid_bf; bf16* h1p_bf;   // [4][R][128]
    bf16* hz_bf;                   // [4][R][kx]
    bf16* z_bf;                    // [R][kz]
    unsigned long long* ts;        // measurement aid (knobs txt_ts_lo / txt_ts_hi, set by the launcher): s_memrealtime stamps of workgroup 0's phases, or null
};
int launch_text_decoder_fwd(const TextDecArgs& a, hipStream_t s);

struct TextDecBwdArgs {
    TextDecArgs f;
    const float* dwords;           // [R][4][12] grad wrt log-probs
    int R_active;                  // rows [0, R_active) carry gradient (text-only pass with lambda 0 never happens here)
    float* dz;                     // [R][D] out
    bf16* dgi0; bf16* dgh0; bf16* dgi1; bf16* dgh1;   // [4][R][304]
    bf16* dlogit_bf;               // [4][R][16]
    bf16* dhinit_bf;               // [R][112]
    float* g_embed;                // [12][100] +=
    float* g_b[4];                 // bih0, bhh0, bih1, bhh1  [300] +=
    float* g_h2o_bias;             // [12] +=
    float* g_z2h_bias;             // [100] +=
};
int launch_text_decoder_bwd(const TextDecBwdArgs& a, hipStream_t s);
