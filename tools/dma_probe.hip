// Probe of the LDS-DMA semantics wgrad_ring.hip relies on (run on the GPU box: hipcc --offload-arch=gfx950 tools/dma_probe.hip -o /tmp/p && /tmp/p):
//   1. `buffer_load_dwordx4 ... offen lds` writes lane l's 16 bytes to M0 + 16*l (M0 a plain LDS byte address, also above 64 KB);
//   2. a lane whose offset fails the descriptor's range check gets ZEROS written (not a skipped write);
//   3. the range check of a raw buffer looks at voffset (+ immediate) only: an in-range voffset with a large soffset still loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma_1k(const __amdgpu_buffer_rsrc_t r, const unsigned lds_addr, const int voff, const int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__global__ void probe(const float* src, int src_bytes, float* out, int lds_base, int soff) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x;
    float* f = reinterpret_cast<float*>(smem + lds_base);
    for (int i = 0; i < 4; ++i) f[lane * 4 + i] = -7.0f;                     // poison
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    // even lanes: reversed in-range source; odd lanes: out of range
    const int vo = (lane & 1) ? (int)0x80000000 : (63 - lane) * 16;
    dma_1k(r, lds0 + lds_base, vo, soff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = f[lane * 4 + i];
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad = 0;
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int lds_base = cfg == 1 ? 100 * 1024 : 2048;                   // above 64 KB in configuration 1
        const int soff = cfg == 2 ? (n * 4 - 4096) : 0;                      // soffset near the end of the buffer in configuration 2
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d, n * 4, o, lds_base, soff);
        float r[256];
        if (hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 2; }
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const float want = (l & 1) ? 0.f : (float)(soff / 4 + (63 - l) * 4 + i);
                if (r[l * 4 + i] != want) { if (bad < 8) printf("cfg %d lane %d elem %d: got %g want %g\n", cfg, l, i, r[l * 4 + i], want); ++bad; }
            }
    }
    printf(bad ? "DMA PROBE FAILED (%d)\n" : "dma probe ok\n", bad);
    return bad ? 1 : 0;
}
