#include "common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

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
Knob g_knobs[96];
int g_nknobs = 0;
std::mutex g_knob_mu;
}
std::atomic<unsigned> g_mmvae_knob_gen{0};
int mmvae_knob_lookup(const char* key, int dflt) {
    std::lock_guard<std::mutex> g(g_knob_mu);
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].key, key) == 0) return g_knobs[i].value;
    return dflt;
}
extern "C" int mmvae_debug_set(const char* key, int value) {
    if (!key || strlen(key) >= sizeof(g_knobs[0].key)) { mmvae_set_error("debug_set: bad key"); return MMVAE_EINVAL; }
    std::lock_guard<std::mutex> g(g_knob_mu);
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].key, key) == 0) { g_knobs[i].value = value; g_mmvae_knob_gen.fetch_add(1); return MMVAE_OK; }
    if (g_nknobs == (int)(sizeof(g_knobs) / sizeof(g_knobs[0]))) { mmvae_set_error("debug_set: table full"); return MMVAE_ENOSPC; }
    strcpy(g_knobs[g_nknobs].key, key);
    g_knobs[g_nknobs++].value = value;
    g_mmvae_knob_gen.fetch_add(1);
    return MMVAE_OK;
}

int mmvae_cu_count() {
    static std::atomic<int> cached[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int n = cached[dev].load();
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 1;
        cached[dev].store(n);
    }
    return n;
}
bool mmvae_serial() {
    static const bool serial = getenv("MMVAE_SERIAL") != nullptr;
    return serial;
}

// ---- completion event of the next launch (common.h: MMVAE_LAUNCH)
static thread_local hipEvent_t g_stop_event = nullptr;
void mmvae_arm_stop_event(hipEvent_t e) { g_stop_event = e; }
hipEvent_t mmvae_take_stop_event() {
    hipEvent_t e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}

// ---- in-step kernel timing (common.h: MMVAE_LAUNCH; include/mmvae_hip.h: mmvae_debug_probe / mmvae_debug_probe_read)
#include <string>
#include <vector>
namespace {
struct ProbeRec { std::string tag, kernel; double flops; hipEvent_t start, stop; };
std::atomic<bool> g_probe{false};
std::mutex g_probe_mu;
std::vector<ProbeRec> g_probe_recs;
std::vector<hipEvent_t> g_probe_pool;
thread_local char g_probe_tag[96] = "";
thread_local double g_probe_flops = 0.0;
}
bool mmvae_probe_on() { return g_probe.load(std::memory_order_relaxed); }
void mmvae_probe_tag(const char* tag, double algo_flops) {
    if (!mmvae_probe_on()) return;
    snprintf(g_probe_tag, sizeof(g_probe_tag), "%s", tag);
    g_probe_flops = algo_flops;
}
void mmvae_probe_events(const char* kernel, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> g(g_probe_mu);
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
        if (!g_probe_pool.empty()) { e[i] = g_probe_pool.back(); g_probe_pool.pop_back(); }
        else if (hipEventCreate(&e[i]) != hipSuccess) { *start = nullptr; *stop = nullptr; return; }
    }
    g_probe_recs.push_back(ProbeRec{g_probe_tag, kernel, g_probe_flops, e[0], e[1]});
    g_probe_tag[0] = 0;
    g_probe_flops = 0.0;
    *start = e[0];
    *stop = e[1];
}
extern "C" int mmvae_debug_probe(int on) {
    g_probe.store(on != 0);
    return MMVAE_OK;
}
// Waits for the probed launches, writes one line per launch "tag\tkernel\tmicroseconds\tflops\n" into buf (truncated at a line
// boundary when cap is too small) and forgets them.  Returns the number of launches written, or a negative error.
extern "C" int mmvae_debug_probe_read(char* buf, long long cap) {
    std::lock_guard<std::mutex> g(g_probe_mu);
    long long used = 0;
    int n = 0;
    int rc = MMVAE_OK;
    for (ProbeRec& r : g_probe_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) {
            mmvae_set_error("probe_read: %s", hipGetErrorString(hipGetLastError()));
            rc = MMVAE_EHIP;
        } else if (buf) {
            char line[512];
            const int len = snprintf(line, sizeof(line), "%s\t%.200s\t%.3f\t%.6g\n", r.tag.c_str(), r.kernel.c_str(), ms * 1e3, r.flops);
            if (used + len < cap) { memcpy(buf + used, line, len); used += len; ++n; }
        }
        g_probe_pool.push_back(r.start);
        g_probe_pool.push_back(r.stop);
    }
    g_probe_recs.clear();
    if (buf && cap > 0) buf[used] = 0;
    return rc != MMVAE_OK ? rc : n;
}
