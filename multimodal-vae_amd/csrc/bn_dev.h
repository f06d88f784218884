// Device helpers shared by the kernels that finalise a BatchNorm layer themselves (elementwise.hip: bn_act_kernel;
// thin.hip: the fused last decoder layer): per-(group, channel) tables from the slot-replicated statistics and the
// running-statistics update (torch.nn.BatchNorm semantics: momentum 0.1, unbiased variance, once per forward call).
#pragma once
#include "elementwise.h"

namespace {

__device__ __forceinline__ void bn_channel_tables(const BnFinalizeArgs& a, int g, int c, float2& aff, float2& mr) {
    const float gamma = a.gamma[c], beta = a.beta[c];
    float mean, rstd;
    if (a.training) {
        float2 s = make_float2(0.f, 0.f);
        for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
            float2 t = a.stats[(g * MMVAE_STAT_SLOTS + q) * a.C + c];
            s.x += t.x; s.y += t.y;
        }
        mean = s.x / a.count;
        float var = fmaxf(s.y / a.count - mean * mean, 0.f);
        rstd = rsqrtf(var + a.eps);
    } else {
        mean = a.running_mean[c];
        rstd = rsqrtf(a.running_var[c] + a.eps);
    }
    aff = make_float2(gamma * rstd, beta - mean * gamma * rstd);
    mr = make_float2(mean, rstd);
}

// running statistics: momentum update once per forward call, groups in pass order; call from ONE workgroup
__device__ __forceinline__ void bn_running_update(const BnFinalizeArgs& f, int tid, int nthreads) {
    if (!f.training || !f.running_mean) return;
    for (int c = tid; c < f.C; c += nthreads) {
        float rm = f.running_mean[c], rv = f.running_var[c];
        for (int g = 0; g < f.G; ++g) {
            float2 s = make_float2(0.f, 0.f);
            for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
                float2 t = f.stats[(g * MMVAE_STAT_SLOTS + q) * f.C + c];
                s.x += t.x; s.y += t.y;
            }
            float mean = s.x / f.count;
            float var = fmaxf(s.y / f.count - mean * mean, 0.f);
            float unbiased = var * f.count / (f.count - 1.f);
            const int nu = ((f.skip_update_mask >> g) & 1u) ? 0 : f.updates_per_group;
            for (int u = 0; u < nu; ++u) {
                rm = (1.f - f.momentum) * rm + f.momentum * mean;
                rv = (1.f - f.momentum) * rv + f.momentum * unbiased;
            }
        }
        f.running_mean[c] = rm; f.running_var[c] = rv;
    }
    if (tid == 0 && f.num_batches_tracked)
        *f.num_batches_tracked += (long long)(f.G - __popc(f.skip_update_mask & ((1u << f.G) - 1u))) * f.updates_per_group;
}

}  // namespace
