#include "common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void mmvae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* mmvae_error_string() { return g_err; }

int mmvae_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mmvae_set_error("%s: %s", what, hipGetErrorString(e));
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}

// ---- measurement aid: MFMA GEMM work (2*rows*N*K per launch, padded taps included) enqueued by this process
static std::atomic<long long> g_mflops{0};
void mmvae_count_flops(double f) { g_mflops.fetch_add((long long)(f * 1e-6)); }
extern "C" double mmvae_debug_flops(int reset) {
    const long long v = reset ? g_mflops.exchange(0) : g_mflops.load();
    return (double)v * 1e6;
}

// ---- side-stream priority policy
#include <atomic>
static std::atomic<int> g_policy{-1};
static std::atomic<bool> g_policy_frozen{false};
int mmvae_stream_policy() { return g_policy.load(); }
void mmvae_stream_policy_freeze() { g_policy_frozen.store(true); }
extern "C" int mmvae_set_stream_policy(int flat) {
    const int want = flat ? 1 : 0;
    if (g_policy_frozen.load()) {
        const int have = g_policy.load() == 0 ? 0 : 1;          // unset = flat
        if (have != want) {
            mmvae_set_error("stream policy: the side streams already exist with policy %d", have);
            return MMVAE_ESTATE;
        }
        return MMVAE_OK;
    }
    g_policy.store(want);
    return MMVAE_OK;
}
