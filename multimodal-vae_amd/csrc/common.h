// Shared device/host helpers for libmmvae_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define MMVAE_OK 0
#define MMVAE_EINVAL (-1)
#define MMVAE_EHIP (-2)
#define MMVAE_ENOSPC (-3)
#define MMVAE_ESTATE (-4)
#define MMVAE_ETIMEOUT (-5)

// thread-local last error (include/mmvae_hip.h: mmvae_last_error)
void mmvae_set_error(const char* fmt, ...);
int mmvae_check_launch(const char* what);
// measurement aid (mmvae_debug_flops): executed GEMM work, counted by the launchers on the host
void mmvae_count_flops(double flops);
// side-stream priority policy (include/mmvae_hip.h: mmvae_set_stream_policy); -1 until somebody asked
int mmvae_stream_policy();
void mmvae_stream_policy_freeze();      // the side streams exist from here on
// test / A-B knobs set through mmvae_debug_set (include/mmvae_hip.h); `dflt` when the key was never set.  The table lives in
// util.cpp behind a mutex; a launcher's call site keeps the value it looked up until the next mmvae_debug_set (a generation
// counter), so the launch path pays one atomic load and a compare per knob, not a lock and a string scan.
#include <atomic>
int mmvae_knob_lookup(const char* key, int dflt);
extern std::atomic<unsigned> g_mmvae_knob_gen;
struct MmvaeKnobSite { std::atomic<unsigned> gen{0xffffffffu}; std::atomic<int> val{0}; };
static inline int mmvae_knob_cached(MmvaeKnobSite& s, const char* key, int dflt) {
    const unsigned g = g_mmvae_knob_gen.load(std::memory_order_acquire);
    if (s.gen.load(std::memory_order_acquire) == g) return s.val.load(std::memory_order_relaxed);
    const int v = mmvae_knob_lookup(key, dflt);
    s.val.store(v, std::memory_order_relaxed);
    s.gen.store(g, std::memory_order_release);
    return v;
}
#define mmvae_knob(key, dflt) ([&]() -> int { static MmvaeKnobSite mmvae_site_; return mmvae_knob_cached(mmvae_site_, key, dflt); }())

#define MMVAE_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            mmvae_set_error(__VA_ARGS__);             \
            return MMVAE_EINVAL;                      \
        }                                             \
    } while (0)

#define MMVAE_TRY(expr)                               \
    do {                                              \
        int _rc = (expr);                             \
        if (_rc != MMVAE_OK) return _rc;              \
    } while (0)

// ---- kernel launch that can carry a completion event
// A fork of the step (side stream waits for the main chain) used to be hipEventRecord on the main stream + hipStreamWaitEvent on
// the side stream.  The record is a barrier packet in the main stream's queue: ~5 us of dispatch latency in front of the NEXT
// main-chain kernel, 14 times per step (measured: 90 us of a 620 us chain).  Bound to the producer kernel's own completion
// signal (hipExtLaunchKernelGGL's stop event) the same event costs the main stream nothing.
//   mmvae_arm_stop_event(e): the next MMVAE_LAUNCH of this thread records `e` at its kernel's completion
//   mmvae_take_stop_event(): used by MMVAE_LAUNCH; returns the armed event (once) or null
#include <hip/hip_ext.h>
void mmvae_arm_stop_event(hipEvent_t e);
hipEvent_t mmvae_take_stop_event();
// In-step kernel timing (mmvae_debug_probe, include/mmvae_hip.h): while the probe is on every MMVAE_LAUNCH carries a start and a
// stop event of its own, labelled with the tag (and algorithmic FLOP count) the launcher set just before with mmvae_probe_tag.
// MMVAE_SERIAL=1 (read once per process): every step on one stream, no overlap -- the profiling aid of tools/prof_serial.sh
bool mmvae_serial();
// compute units of the current device (cached per device): co-residency limits of the cluster kernels derive from it
int mmvae_cu_count();
bool mmvae_probe_on();
void mmvae_probe_tag(const char* tag, double algo_flops);
void mmvae_probe_events(const char* kernel, hipEvent_t* start, hipEvent_t* stop);
#define MMVAE_LAUNCH(kernel, grid, block, lds, stream, ...)                                              \
    do {                                                                                                 \
        hipEvent_t _se = mmvae_take_stop_event(), _st = nullptr, _fork = nullptr;                        \
        if (mmvae_probe_on()) { _fork = _se; _se = nullptr; mmvae_probe_events(#kernel, &_st, &_se); }   \
        if (_se) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _st, _se, 0, __VA_ARGS__);      \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                          \
        if (_fork) (void)hipEventRecord(_fork, stream);                                                  \
    } while (0)

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2 };

// Same-address float atomics serialise at the memory side (MI355X_MICROARCH.md, Global float atomics: one row
// shared by every workgroup is ~14x slower), so every per-channel reduction target is replicated in SLOTS copies
// picked by the workgroup id; the (tiny) consumer sums the copies.
#define MMVAE_STAT_SLOTS 16
#define MMVAE_LOSS_SLOTS 32

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float act_fwd(int act, float x) {
    if (act == ACT_SWISH) return x * sigmoidf_(x);
    if (act == ACT_RELU) return x > 0.f ? x : 0.f;
    return x;
}
// derivative of act at pre-activation x
__device__ __forceinline__ float act_bwd(int act, float x) {
    if (act == ACT_SWISH) {
        float s = sigmoidf_(x);
        return s * (1.0f + x * (1.0f - s));
    }
    if (act == ACT_RELU) return x > 0.f ? 1.f : 0.f;
    return 1.f;
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- Philox4x32-10 counter RNG (production eps / dropout masks) ----
struct Philox {
    __device__ static inline uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
    __device__ static inline void round(uint32_t (&c)[4], uint32_t (&k)[2]) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t hi0 = mulhi(M0, c[0]), lo0 = M0 * c[0];
        uint32_t hi1 = mulhi(M1, c[2]), lo1 = M1 * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    __device__ static inline void gen(uint64_t seed, uint64_t ctr, uint32_t stream, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0x5bd1e995u};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
        for (int i = 0; i < 10; ++i) round(c, k);
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// One-time-per-DEVICE set-up (kernel attributes such as dynamic LDS above 64 KB): the bit mask is keyed by the current
// device, so a process that drives several devices repeats the set-up on each (no "first device wins" global state).
#include <atomic>
static inline bool mmvae_first_use_on_device(std::atomic<unsigned>& mask) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned bit = 1u << (d & 31);
    return (mask.fetch_or(bit) & bit) == 0;
}

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
