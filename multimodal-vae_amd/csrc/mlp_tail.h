// Fused row-block launches for the last two Linears of the image encoder's classifier (n_latents = 100). See mlp_tail.hip.
#pragma once
#include "common.h"

struct Mlp2FwdArgs {
    int rows;
    const bf16* x;           // [rows][400] activated (+ dropout) output of classifier.0
    const bf16* w2;          // classifier.3 weight, fragment-major [208][416]
    const float* b2;
    const bf16* w3;          // classifier.6 weight, fragment-major [208][224]
    const float* b3;
    const uint8_t* mask; float mask_scale;      // [rows][200] keep flags of the second Dropout, or null
    bf16 *y2, *ay2;          // out [rows][200]: raw and activated (+ dropout) classifier.3 output (saved for the backward pass)
    float* out;              // out [rows][200] = (mu | logvar)
};
int launch_mlp2_fwd(const Mlp2FwdArgs& a, hipStream_t s);

struct Mlp2BwdArgs {
    int rows;
    const bf16* d_out;       // [rows][200] gradient wrt (mu | logvar)
    const bf16* w3t;         // classifier.6 weight transposed, fragment-major [208][224]  (row = classifier.3 unit)
    const bf16* w2t;         // classifier.3 weight transposed, fragment-major [400][224]  (row = classifier.0 unit)
    const bf16 *y2, *y1;     // raw pre-activations [rows][200], [rows][400]
    const uint8_t *mask2, *mask1; float mask_scale;
    bf16 *dy2, *dy1;         // out: gradients wrt the raw outputs of classifier.3 / classifier.0
    float *db2, *db1;        // += column sums of dy2 / dy1 (bias gradients of classifier.3 / classifier.0)
};
int launch_mlp2_bwd(const Mlp2BwdArgs& a, hipStream_t s);
