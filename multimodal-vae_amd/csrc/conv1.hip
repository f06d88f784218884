// First layer of the MultiMNIST image encoder on the matrix cores: Conv2d(1, 32, 4, 2, 1) on a 50x50 image + Swish, and its
// weight gradient (multimnist/model.py:160-161).
//
// Before: im2col_small_kernel (fp32 image -> bf16 patches in HBM) + a gather GEMM with K = 16 (10 TFLOP/s: 16 of the 64 k
// columns of a tile are real) forward, and a streamed weight gradient over the same patches as the LAST kernel of the backward
// chain -- 41 us of the main chain for 0.5 GFLOP.  Here one workgroup owns one image: the image sits in LDS as bf16 with a zero
// halo, the 625 x 16 patch matrix is built there once, and
//   forward      r1[pixel][32] = patch[pixel][16 taps] . W[16 taps][32]     one 32x32x16 MFMA per 32 pixels; raw and Swish
//                                                                            copies leave through the conv kernels' epilogue
//   weight grad  dW[32][16 taps] = d1[pixel][32]^T . patch[pixel][16]       transposed LDS reads, k = 640 pixel rows over the
//                                                                            waves, per-image partial added with float atomics
//                                                                            (256 x 512 adds: no slab, no reduce launch on the tail)
#include "thin.h"
#include "convres_epi.h"

namespace {

constexpr int C1_IMG = 50, C1_OH = 25, C1_NPIX = 625, C1_TILES = 20, C1_ROWS = 640, C1_C = 32;
constexpr int C1_HW = 52;              // image rows / columns with a zero halo
constexpr int C1_PP = 48;              // patches: bytes per pixel (16 bf16 + 16)
constexpr int C1_AP = 80;              // gradient tile: bytes per pixel (32 bf16 + 16)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4e;
__device__ __forceinline__ bf16x8 tr_pair_e(const char* a0, const char* a1) {
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4e*)a0);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4e*)a1);
    return u.v;
}

// image (fp32, global) -> bf16 with halo in LDS; patches[pixel][tap] = image(2oy-1+kh, 2ox-1+kw) (padding rows zero)
template <int NTHR>
__device__ __forceinline__ void c1_patches(const float* img, bf16* img_s, char* pat_s, int tid) {
    for (int i = tid; i < C1_HW * C1_HW; i += NTHR) {
        const int y = i / C1_HW - 1, x = i - (y + 1) * C1_HW - 1;
        img_s[i] = ((unsigned)y < (unsigned)C1_IMG && (unsigned)x < (unsigned)C1_IMG) ? (bf16)img[y * C1_IMG + x] : (bf16)0.f;
    }
    __syncthreads();
    for (int px = tid; px < C1_ROWS; px += NTHR) {
        bf16x8 lo, hi;
        if (px < C1_NPIX) {
            const int oy = px / C1_OH, ox = px - oy * C1_OH;
            const bf16* d0 = img_s + (2 * oy) * C1_HW + 2 * ox;          // (2oy-1+kh) + 1 halo row, (2ox-1+kw) + 1 halo column
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                lo[kw] = d0[kw]; lo[4 + kw] = d0[C1_HW + kw];
                hi[kw] = d0[2 * C1_HW + kw]; hi[4 + kw] = d0[3 * C1_HW + kw];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { lo[j] = (bf16)0.f; hi[j] = (bf16)0.f; }
        }
        *reinterpret_cast<bf16x8*>(pat_s + px * C1_PP) = lo;
        *reinterpret_cast<bf16x8*>(pat_s + px * C1_PP + 16) = hi;
    }
    __syncthreads();
}

struct Conv1FwdArgs { const float* image; int B; const bf16* Wp; int Kpad; bf16* r1; bf16* a1; };

constexpr int C1F_WAVES = 4, C1F_NTHR = C1F_WAVES * 64;
constexpr int C1F_OFF_IMG = 0, C1F_OFF_PAT = C1F_OFF_IMG + ((C1_HW * C1_HW * 2 + 15) / 16) * 16, C1F_OFF_SCR = C1F_OFF_PAT + C1_ROWS * C1_PP,
              C1F_LDS = C1F_OFF_SCR + C1F_WAVES * 2560;

__global__ __launch_bounds__(C1F_NTHR) void conv1_fwd_mfma_kernel(const Conv1FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    c1_patches<C1F_NTHR>(a.image + (size_t)n * (C1_IMG * C1_IMG), reinterpret_cast<bf16*>(smem + C1F_OFF_IMG), smem + C1F_OFF_PAT, tid);
    // B[k = tap][j = channel]: packed weights [32][Kpad], k = tap
    const bf16x8 wb = *reinterpret_cast<const bf16x8*>(a.Wp + (size_t)r * a.Kpad + 8 * h);
    const size_t bytes = (size_t)a.B * C1_NPIX * C1_C * 2;
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(a.r1, 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(a.a1, 0, (int)bytes, 0x00020000);
    char* const scr = smem + C1F_OFF_SCR + wave * 2560;
    float s1 = 0.f, s2 = 0.f;
    for (int t = wave; t < C1_TILES; t += C1F_WAVES) {
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const bf16x8 pa = *reinterpret_cast<const bf16x8*>(smem + C1F_OFF_PAT + (t * 32 + r) * C1_PP + h * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, wb, acc, 0, 0, 0);
        const unsigned base = (unsigned)(((size_t)n * C1_NPIX + t * 32) * (C1_C * 2)) + (unsigned)((lane >> 4) * 16);
        const int p0 = t * 32 + (lane & 15), p1 = p0 + 16;
        const unsigned off0 = p0 < C1_NPIX ? base + (unsigned)((lane & 15) * (C1_C * 2)) : 0x40000000u;
        const unsigned off1 = p1 < C1_NPIX ? base + (unsigned)((16 + (lane & 15)) * (C1_C * 2)) : 0x40000000u;
        f32x16 act;
#pragma unroll
        for (int j = 0; j < 16; ++j) act[j] = swish_fast(acc[j]);
        // (rows past the image carry an out-of-range offset: dropped; the sums of the epilogue are not used here)
        cr_epilogue_tile<0, false>(acc, scr, lane, 32, off0, off1, rsrc_r, rsrc_r, 1.f, 0.f, 0.f, 0.f, s1, s2, true);
        cr_epilogue_tile<0, false>(act, scr, lane, 32, off0, off1, rsrc_a, rsrc_a, 1.f, 0.f, 0.f, 0.f, s1, s2, true);
    }
}

struct Conv1WgradArgs { const float* image; int B; const bf16* d1; float* dWp; int Kpad; };

constexpr int C1W_WAVES = 8, C1W_NTHR = C1W_WAVES * 64;
constexpr int C1W_OFF_D = 0, C1W_OFF_IMG = C1W_OFF_D + C1_ROWS * C1_AP, C1W_OFF_PAT = C1W_OFF_IMG + ((C1_HW * C1_HW * 2 + 15) / 16) * 16,
              C1W_OFF_RED = C1W_OFF_PAT + C1_ROWS * C1_PP, C1W_LDS = C1W_OFF_RED + C1W_WAVES * C1_C * 16 * 4;

__global__ __launch_bounds__(C1W_NTHR) void conv1_wgrad_mfma_kernel(const Conv1WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x;
    char* const D_s = smem + C1W_OFF_D;
    // gradient image [625][32] -> LDS rows of 80 bytes (padding rows zero)
    {
        constexpr int NV = C1_NPIX * 4;
        const bf16* src = a.d1 + (size_t)n * C1_NPIX * C1_C;
        for (int v = tid; v < C1_ROWS * 4; v += C1W_NTHR) {
            i32x4c val = {0, 0, 0, 0};
            if (v < NV) val = *reinterpret_cast<const i32x4c*>(src + (size_t)v * 8);
            *reinterpret_cast<i32x4c*>(D_s + (v >> 2) * C1_AP + (v & 3) * 16) = val;
        }
    }
    c1_patches<C1W_NTHR>(a.image + (size_t)n * (C1_IMG * C1_IMG), reinterpret_cast<bf16*>(smem + C1W_OFF_IMG), smem + C1W_OFF_PAT, tid);
    // dW[ch][tap] = sum over pixel rows: A[i = ch][k = pixel] from the gradient tile, B[k = pixel][j = tap] from the patches
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int ks = wave; ks < C1_ROWS / 16; ks += C1W_WAVES) {
        const int row = ks * 16 + 8 * h + q;
        const char* a0 = D_s + row * C1_AP + (16 * (g4 & 1) + 4 * p) * 2;
        const char* b0 = smem + C1W_OFF_PAT + row * C1_PP + (4 * p) * 2;          // taps 0..15 for both column halves (16..31 unused)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pair_e(a0, a0 + 4 * C1_AP), tr_pair_e(b0, b0 + 4 * C1_PP), acc, 0, 0, 0);
    }
    float* wred = reinterpret_cast<float*>(smem + C1W_OFF_RED) + wave * (C1_C * 16);
    if (r < 16) {                                            // lane = tap, register j = channel (j&3) + 8*(j>>2) + 4h
#pragma unroll
        for (int j = 0; j < 16; ++j) wred[((j & 3) + 8 * (j >> 2) + 4 * h) * 16 + r] = acc[j];
    }
    __syncthreads();
    {
        const float* wr = reinterpret_cast<const float*>(smem + C1W_OFF_RED);
        for (int i = tid; i < C1_C * 16; i += C1W_NTHR) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < C1W_WAVES; ++w) s += wr[w * C1_C * 16 + i];
            atomicAdd(a.dWp + (size_t)(i >> 4) * a.Kpad + (i & 15), s);        // packed gradient [channel][Kpad], k = tap
        }
    }
}

}  // namespace

int launch_conv1_fwd_mfma(const float* image, int B, const bf16* Wp, int Kpad, bf16* r1, bf16* a1, hipStream_t s) {
    MMVAE_REQUIRE(image && Wp && r1 && a1 && Kpad >= 16 && (size_t)B * C1_NPIX * C1_C * 2 < 0x40000000ull, "conv1 fwd: arguments");
    Conv1FwdArgs a{image, B, Wp, Kpad, r1, a1};
    MMVAE_LAUNCH(conv1_fwd_mfma_kernel, dim3(B), dim3(C1F_NTHR), C1F_LDS, s, a);
    return mmvae_check_launch("conv1_fwd_mfma");
}
int launch_conv1_wgrad_mfma(const float* image, int B, const bf16* d1, float* dWp, int Kpad, hipStream_t s) {
    MMVAE_REQUIRE(image && d1 && dWp && Kpad >= 16, "conv1 wgrad: arguments");
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_wgrad_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C1W_LDS);
    Conv1WgradArgs a{image, B, d1, dWp, Kpad};
    MMVAE_LAUNCH(conv1_wgrad_mfma_kernel, dim3(B), dim3(C1W_NTHR), C1W_LDS, s, a);
    return mmvae_check_launch("conv1_wgrad_mfma");
}
