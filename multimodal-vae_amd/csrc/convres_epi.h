// Device pieces shared by the image-resident kernels (convres.hip, dec_last.hip): fast Swish, and the epilogue that takes a
// 32x32 accumulator tile to 16-byte global stores through a wave-private LDS scratch.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4c;

// derivative of Swish at pre-activation x (hardware exp / rcp: the result multiplies a bf16-rounded gradient)
__device__ __forceinline__ float dswish_fast(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float swish_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4c;

// Epilogue of one 32x32 accumulator tile (lane = channel c = lane&31, register j = row (j&3) + 8*(j>>2) + 4*(lane>>5)).
// The tile crosses a wave-private LDS scratch so that every global access is a 16-byte vector of 8 channels of one pixel
// (2-byte accesses per lane cost one address per lane in the texture path: the first version, 16 short stores per tile,
// spent as long storing as computing):
//   * results leave as [channel][32 rows] (4 packed 8-byte writes per lane) and come back through ds_read_b64_tr_b16:
//     lane (g = lane>>4, i = lane&15) gets channels 8g..8g+7 of rows i and 16+i -- its two 16-byte stores;
//   * MODE 1: the saved tensor of the output geometry arrives the other way round: two 16-byte loads per lane in that same
//     (row, channel octet) mapping, written as [row][32 channels], read back transposed into the accumulator layout.
// `off0` / `off1`: byte offsets (row table + channel octet + image base) of this lane's two vectors; rows >= rows_valid
// (PARTIAL tiles only) carry an out-of-range offset (dropped by the buffer range check) and are kept out of the sums.
template <int MODE, bool PARTIAL>
__device__ __forceinline__ void cr_epilogue_tile(const f32x16& acc, char* scr, int lane, int rows_valid, unsigned off0, unsigned off1,
                                                 __amdgpu_buffer_rsrc_t orsrc, __amdgpu_buffer_rsrc_t rrsrc, float dsc, float dsh,
                                                 float dmean, float drstd, float& s1, float& s2, bool store) {
    constexpr int SP = 80;                                       // scratch row pitch: 32 x bf16 + 16
    const int r = lane & 31, h = lane >> 5, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = acc[j];
    if constexpr (MODE == 1) {
        const i32x4c r0 = __builtin_bit_cast(i32x4c, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)off0, 0, 0));
        const i32x4c r1 = __builtin_bit_cast(i32x4c, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)off1, 0, 0));
        *reinterpret_cast<i32x4c*>(scr + i * SP + g * 16) = r0;              // [row][channel]
        *reinterpret_cast<i32x4c*>(scr + (16 + i) * SP + g * 16) = r1;
        asm volatile("" ::: "memory");      // wave-private scratch: LDS order within a wave is issue order
        // block of 4 rows x 16 channels per 16-lane group: lane 4q+p addresses row q, channels 4p..4p+3 and receives channel i
        const char* rb = scr + (4 * h + q) * SP + (16 * ((lane >> 4) & 1) + 4 * p) * 2;
        f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (cast the WHOLE vector: __builtin_bit_cast on an element of an ext-vector returns element 0 for every index)
            const bf16x4 rv = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(rb + 8 * k * SP)));
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const f32x2 rr = {(float)rv[e], (float)rv[e + 1]};
                const f32x2 x = rr * dsc + dsh;
                f32x2 d = {dswish_fast(x[0]), dswish_fast(x[1])};
                f32x2 o = {v[4 * k + e], v[4 * k + e + 1]};
                o *= d;
                if constexpr (PARTIAL) {
                    const int row = 8 * k + 4 * h + e;
                    o[0] = row < rows_valid ? o[0] : 0.f;
                    o[1] = row + 1 < rows_valid ? o[1] : 0.f;
                }
                t1 += o;
                t2 += o * ((rr - dmean) * drstd);
                v[4 * k + e] = o[0]; v[4 * k + e + 1] = o[1];
            }
        }
        s1 += t1[0] + t1[1];
        s2 += t2[0] + t2[1];
        asm volatile("" ::: "memory");
    } else {
        f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            f32x2 o = {v[j], v[j + 1]};
            if constexpr (PARTIAL) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                o[0] = row < rows_valid ? o[0] : 0.f;
                o[1] = row + 1 < rows_valid ? o[1] : 0.f;
                v[j] = o[0]; v[j + 1] = o[1];
            }
            t1 += o;
            t2 += o * o;
        }
        s1 += t1[0] + t1[1];
        s2 += t2[0] + t2[1];
    }
    // results -> [channel][32 rows] bf16: this lane's 4 runs of 4 rows
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[4 * k + e];
        *reinterpret_cast<bf16x4*>(scr + r * SP + (8 * k + 4 * h) * 2) = o;
    }
    asm volatile("" ::: "memory");
    if (store) {
        // block of 4 channels x 16 rows: lane 4q+p addresses channel c0+q, rows 4p..4p+3 (+16 for the second vector) and
        // receives row i
        const char* ob = scr + (8 * g + q) * SP + (4 * p) * 2;
        union { struct { s16x4 a, b; } s; i32x4c v; } u0, u1;
        u0.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(ob));
        u0.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(ob + 4 * SP));
        u1.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(ob + 32));
        u1.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)(ob + 4 * SP + 32));
        __builtin_amdgcn_raw_buffer_store_b128(u0.v, orsrc, (int)off0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u1.v, orsrc, (int)off1, 0, 0);
    }
    asm volatile("" ::: "memory");
}

}  // namespace
