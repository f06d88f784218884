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

// ---- test / A-B knobs (include/mmvae_hip.h: mmvae_debug_set): a small process-wide table of named integers
#include <mutex>
#include <cstring>
namespace {
struct Knob { char key[32]; int value; };
Knob g_knobs[32];
int g_nknobs = 0;
std::mutex g_knob_mu;
}
int mmvae_knob(const char* key, int dflt) {
    std::lock_guard<std::mutex> g(g_knob_mu);
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].key, key) == 0) return g_knobs[i].value;
    return dflt;
}
extern "C" int mmvae_debug_set(const char* key, int value) {
    if (!key || strlen(key) >= sizeof(g_knobs[0].key)) { mmvae_set_error("debug_set: bad key"); return MMVAE_EINVAL; }
    std::lock_guard<std::mutex> g(g_knob_mu);
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].key, key) == 0) { g_knobs[i].value = value; return MMVAE_OK; }
    if (g_nknobs == 32) { mmvae_set_error("debug_set: table full"); return MMVAE_ENOSPC; }
    strcpy(g_knobs[g_nknobs].key, key);
    g_knobs[g_nknobs++].value = value;
    return MMVAE_OK;
}

// ---- completion event of the next launch (common.h: MMVAE_LAUNCH)
static thread_local hipEvent_t g_stop_event = nullptr;
void mmvae_arm_stop_event(hipEvent_t e) { g_stop_event = e; }
hipEvent_t mmvae_take_stop_event() {
    hipEvent_t e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}
