// Probe: what does a stream fork cost the MAIN chain, by mechanism?  (hipcc --offload-arch=gfx950 tools/fork_probe.hip -o /tmp/fork_probe)
//   main stream:  [A_i, B_i] x N     side stream: S_i behind A_i
//   mode 0  no side work at all
//   mode 1  hipEventRecord after A_i + hipStreamWaitEvent on the side stream
//   mode 2  A_i launched with a completion (stop) event, hipStreamWaitEvent on the side stream   (what plan_base.h does)
//   mode 3  A_i's last workgroup writes i+1 to a flag word; side stream: hipStreamWaitValue32(flag >= i+1)   (no packet on main)
//   mode 4  as 3 but the flag is written by hipStreamWriteValue32 on the main stream
// S_i checks that A_i's data is complete (counts errors).  Prints us per (A, B) pair on the main stream and the errors.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void kA(float* data, int n, int seq, unsigned* ticket, unsigned* flag, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = (float)seq;
    for (int k = 0; k < spin; ++k) v = v * 1.0000001f + 1e-9f;
    if (i < n) data[i] = (float)seq + (v - v);
    if (flag) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = atomicAdd(ticket, 1u);
            if (t == gridDim.x - 1) { *ticket = 0u; __hip_atomic_store(flag, (unsigned)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}
__global__ void kB(float* out, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 1.f;
    for (int k = 0; k < spin; ++k) v = v * 1.0000001f + 1e-9f;
    if (i < n) out[i] = v;
}
__global__ void kS(const float* data, int n, int seq, unsigned* errors) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && data[i] < (float)seq) atomicAdd(errors, 1u);      // data[i] >= seq: A_seq (or a later A) has written it
}

int main(int argc, char** argv) {
    const int N = 200, n = 256 * 256, spin = argc > 1 ? atoi(argv[1]) : 2000;
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    float *data, *out; unsigned *ticket, *errors, *flag_dev, *flag_sig = nullptr, *flag_host;
    CK(hipMalloc(&data, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&flag_dev, 8));
    if (hipExtMallocWithFlags((void**)&flag_sig, 8, hipMallocSignalMemory) != hipSuccess) { flag_sig = nullptr; (void)hipGetLastError(); printf("no signal memory\n"); }
    CK(hipHostMalloc((void**)&flag_host, 8, hipHostMallocDefault));
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    struct Mode { int mode; unsigned* flag; const char* name; };
    std::vector<Mode> modes = {{0, nullptr, "no side work"}, {1, nullptr, "event record + wait"}, {2, nullptr, "completion event + wait"},
                               {3, flag_dev, "kernel flag (device memory) + WaitValue32"}, {3, flag_host, "kernel flag (pinned host memory) + WaitValue32"},
                               {4, flag_dev, "WriteValue32 (device memory) + WaitValue32"}};
    if (flag_sig) { modes.push_back({3, flag_sig, "kernel flag (signal memory) + WaitValue32"}); modes.push_back({4, flag_sig, "WriteValue32 (signal memory) + WaitValue32"}); }
    for (int rep = 0; rep < 2; ++rep)
    for (const Mode& md : modes) {
        CK(hipMemset(data, 0, n * 4)); CK(hipMemset(ticket, 0, 4)); CK(hipMemset(errors, 0, 4));
        if (md.flag) { if (md.flag == flag_host) *flag_host = 0; else CK(hipMemset(md.flag, 0, 8)); }
        CK(hipDeviceSynchronize());
        bool ok = true;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N && ok; ++i) {
            const int seq = i + 1;
            if (md.mode == 2) hipExtLaunchKernelGGL(kA, dim3(256), dim3(256), 0, m, nullptr, ev[i], 0, data, n, seq, ticket, (unsigned*)nullptr, spin);
            else hipLaunchKernelGGL(kA, dim3(256), dim3(256), 0, m, data, n, seq, ticket, md.mode == 3 ? md.flag : nullptr, spin);
            if (md.mode == 1) CK(hipEventRecord(ev[i], m));
            if (md.mode == 4 && hipStreamWriteValue32(m, md.flag, (unsigned)seq, 0) != hipSuccess) { printf("  WriteValue32 failed: %s\n", hipGetErrorString(hipGetLastError())); ok = false; break; }
            hipLaunchKernelGGL(kB, dim3(256), dim3(256), 0, m, out, n, spin);
            if (md.mode == 1 || md.mode == 2) CK(hipStreamWaitEvent(s, ev[i], 0));
            if (md.mode >= 3 && hipStreamWaitValue32(s, md.flag, (unsigned)seq, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
                printf("  WaitValue32 failed: %s\n", hipGetErrorString(hipGetLastError())); ok = false; break;
            }
            if (md.mode != 0) hipLaunchKernelGGL(kS, dim3(256), dim3(256), 0, s, data, n, seq, errors);
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (!ok) { CK(hipDeviceSynchronize()); continue; }
        CK(hipStreamSynchronize(m));
        const auto t2 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        const auto t3 = std::chrono::steady_clock::now();
        unsigned err = 0; CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        if (rep == 1)
            printf("%-52s main chain %7.2f us per pair (enqueue %6.2f)  side drained +%7.1f us  order errors %u\n", md.name, us(t0, t2) / N, us(t0, t1) / N, us(t2, t3), err);
    }
    return 0;
}
