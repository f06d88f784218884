// Ring-staged weight-gradient kernel for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the conv / transposed-conv weight
// gradients of the stride-2 (and stride-1) layers whose images fit LDS.  Geometry and slot layout: wgrad_geo.h.
//
//   dW[n][(ty, tx)][c] = sum over images, pixels of  S[img][oy][ox][n] * B[img][oy*ST - PAD + ty][ox*ST - PAD + tx][c]
//
// What the streamed kernel (gemm.hip: wgrad_kernel) pays for and this one does not:
//   * it gathers the big-side rows once per TAP (im2col through L2: 4.3x the algorithmic bytes on hallucinate.6) -- here a
//     workgroup owns one stride-parity CLASS of taps: it needs the small image and ONE parity plane (a quarter) of the big
//     image per sample, every byte crosses L2 -> LDS once per workgroup, and a tap is an immediate cell offset;
//   * its loads go through registers with per-vector address and bounds arithmetic, one 64-row iteration at a time between
//     two barriers -- here the slots of a ring are filled by LDS-DMA (buffer_load ... lds, 1 KB per wave instruction, the
//     per-lane source offsets computed once per kernel, ring cells and padding rows out of range = hardware zero fill) and
//     stay in flight across the single barrier per batch (counted vmcnt);
//   * 16x16x32 MFMAs on 64x64 wave tiles -- here 32x32x16 with both operands through transposed LDS reads;
//   * one partial tile per 64..128 rows -- here the accumulators of a workgroup's slice (NS x taps of the class x C) live in
//     registers across ALL its images; a class has as many partial copies as it has image groups (chosen per class in
//     proportion to its tap count so that every workgroup carries the same work), summed by one reduce launch.
#include "gemm.h"
#include "wgrad_geo.h"
#include "wgrad_ring.h"
#include "wgrad_ring_geos.h"
#include <type_traits>
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4r;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int WR_MAXCLS = 8;
constexpr int WR_MAXJOBS = 1024;

struct WrArgs {
    const bf16* S;              // small-side tensor [nimg][OH*OW][N]
    const bf16* Bg;             // big-side tensor [nimg][AH][AW][C]
    float* slab;                // class c, group g: slab + slab_off[c] + g * N * Kc(c)   as [N][Kc(c)], k = class tap * C + channel
    long long slab_off[WR_MAXCLS];
    int groups[WR_MAXCLS];      // image groups of each class (a class has MS * groups[c] workgroups)
    unsigned short job[WR_MAXJOBS];     // workgroup -> class << 12 | (group * MS + slice): see try_wr for the order
    int units;                  // batches of IB images in the tensors
    int nimg;
    int dbg;                    // measurement aid: 1 no stores, 2 no MFMA loop, 4 no DMA
    unsigned long long* ts;     // measurement aid (knobs wr_ts_lo / wr_ts_hi): 8 s_memrealtime stamps (100 MHz) per workgroup, or null
    float* atomic_dst;          // non-null: no partial copies -- the accumulators are ADDED (fp32 atomics) into the zeroed packed gradient
    int atomic_kpad;            //           [N][Kpad], k = tap * C + c (layers with a small dW: the adds of all groups are a few MB)
};

// two transposed 8-byte reads = one 32x32x16 operand fragment whose k axis runs over LDS rows
__device__ __forceinline__ bf16x8 tr_pair_r(const char* a0, const char* a1) {
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4r*)a0);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4r*)a1);
    return u.v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int OOB = (int)0x80000000;        // a buffer offset past every tensor: the hardware range check returns zeros

// One LDS-DMA wave instruction: 64 lanes x 16 bytes from (descriptor base + soff + the lane's voff) to LDS bytes
// [lds_addr, lds_addr + 1024).  Inline asm on purpose: hipcc treats the builtin form as a pending LDS write and drains vmcnt(0)
// in front of the next ds_read -- the fills of the ring must stay in flight across the barrier and the MFMAs; the kernel counts
// them itself (wait_vmcnt).  M0 (the LDS base of the DMA) is written in the same statement that uses it.
__device__ __forceinline__ void dma_1k(const __amdgpu_buffer_rsrc_t r, const unsigned lds_addr, const int voff, const int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory");
}

template <class G, int CLS>
__device__ __forceinline__ void wr_body(const WrArgs& a, const int ms, const int u0, const int u1, float* const copy, char* const smem) {
    constexpr int WAVES = G::WAVES, NF = G::NF, SLOTS = G::SLOTS, KST = G::KST, CPW = G::CPW(CLS), NCT = G::NCT(CLS), CT = G::CT;
    constexpr int SLOT = G::SLOT_BYTES, NTN = G::NTN, WC = G::WC, NTX = G::NTX(CLS), LC = G::LC(CLS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long t_start = 0, t_setup = 0, t_first = 0, t_wait = 0, t_loop = 0, t_issue = 0;
    if (a.ts) t_start = __builtin_amdgcn_s_memrealtime();

    // ---- DMA sources of this lane's chunks of a slot (wave instruction j of wave w covers chunks (j*WAVES + w)*64 .. +63)
    int voff[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const wrgeo::Src s = G::src(CLS, ((j * WAVES + wave) * 64) + lane);
        voff[j] = s.tensor < 0 ? OOB : s.off;
    }
    const int n0 = ms * G::NS;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.S + n0), 0, (int)((size_t)a.nimg * G::OYX * G::N * 2 - (size_t)n0 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.Bg), 0, (int)((size_t)a.nimg * G::AH * G::AW * G::C * 2), 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    // pieces [j0, j1) of the fill of batch u into `slot`.  A batch past the group's last one: every lane out of range (the fill
    // keeps the vmcnt bookkeeping uniform and costs no memory traffic); dbg & 4: the same for every batch
    auto issue = [&](int u, int slot, int j0, int j1) {
        const bool live = u < u1 && !(a.dbg & 4);
        const int soff_s = u * (G::IB * G::OYX * G::N * 2), soff_b = u * (G::IB * G::AH * G::AW * G::C * 2);
        const unsigned base = lds0 + slot * SLOT + wave * 1024;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j < j0 || j >= j1) continue;
            const int vo = live ? voff[j] : OOB;
            if (j < G::NFS) dma_1k(rs, base + j * (WAVES * 1024), vo, soff_s);
            else dma_1k(rb, base + j * (WAVES * 1024), vo, soff_b);
        }
    };
#pragma unroll
    for (int s = 0; s < SLOTS - 1; ++s) issue(u0 + s, s, 0, NF);

    // ---- per-lane constants of the transposed reads: lane (h, q, p) addresses k-row 8h + q (and + 4), channels 16*(g&1) + 4p ..
    const int h = lane >> 5, q4 = (lane & 15) >> 2;
    const int lanecol = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int wr = wave % NTN, wc = wave / NTN;                  // this wave's row tile of the slice and its column-tile lane
    const int a_base = G::sm_off(wr, 8 * h + q4) + lanecol;      // + ks*1024 (+256 for the second read)
    int rb0[KST], rb1[KST];                                      // gathered operand: byte offset of the base cell of the lane's k-rows
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
        const int kr = ks * 16 + 8 * h + q4;
        rb0[ks] = G::SM_BYTES + G::rowcell(CLS, kr) * 64 + lanecol;
        rb1[ks] = G::SM_BYTES + G::rowcell(CLS, kr + 4) * 64 + lanecol;
    }
    int coloff[CPW];                                             // per column tile: (channel tile plane, tap cell) byte offset
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int col = wc + j * WC, cc = col < NCT ? col : 0;
        const int k = cc / CT, ct = cc - k * CT;
        coloff[j] = (ct * G::NCELLP(CLS) + (k / NTX) * LC + k % NTX) * 64;
    }
    f32x16 acc[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // ---- main loop: one barrier per batch; the fills of the next SLOTS-1 batches are in flight while this one is multiplied
    int slot = 0;
    if (a.ts) t_setup = __builtin_amdgcn_s_memrealtime();
    for (int u = u0; u < u1; ++u) {
        unsigned long long tw = 0;
        if (a.ts) tw = __builtin_amdgcn_s_memrealtime();
        wait_vmcnt<(SLOTS - 2) * NF>();                          // this wave's part of batch u has landed
        __builtin_amdgcn_s_barrier();                            // ... everybody's; and everybody is done reading batch u-1
        if (a.ts) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); if (u == u0) t_first = t; else t_wait += t - tw; }
        // the fill of batch u+SLOTS-1 goes into the slot batch u-1 was read from.  Its pieces are issued BETWEEN the k-steps
        // (SPREAD): an LDS-DMA instruction holds the issuing wave for 100+ cycles (measured: 0.66 us of a batch's 1.1-1.6 us
        // were the nine pieces in front of the first MFMA), behind a k-step's MFMAs that time runs under the matrix pipe
        int ns = slot + SLOTS - 1; if (ns >= SLOTS) ns -= SLOTS;
        constexpr bool SPREAD = SLOTS >= 3;          // (a two-slot ring needs the fill back by the next barrier: issue it at once)
        unsigned long long ti = 0;
        if (a.ts) ti = __builtin_amdgcn_s_memrealtime();
        if (!SPREAD || (a.dbg & 2)) issue(u + SLOTS - 1, ns, 0, NF);
        if (a.ts) t_issue += __builtin_amdgcn_s_memrealtime() - ti;
        const char* const sb = smem + slot * SLOT;
        if (!(a.dbg & 2)) {
            // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (two register sets): with one wave per SIMD nothing
            // else covers the LDS latency (measured 60 cycles per MFMA with the reads one MFMA ahead, 32 is the pipe's rate)
            bf16x8 af[2], bfr[2][CPW];
            af[0] = tr_pair_r(sb + a_base, sb + a_base + 256);
#pragma unroll
            for (int j = 0; j < CPW; ++j) bfr[0][j] = tr_pair_r(sb + rb0[0] + coloff[j], sb + rb1[0] + coloff[j]);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                if (SPREAD) issue(u + SLOTS - 1, ns, ks * NF / KST, (ks + 1) * NF / KST);
                if (ks + 1 < KST) {
                    af[(ks + 1) & 1] = tr_pair_r(sb + a_base + (ks + 1) * 1024, sb + a_base + (ks + 1) * 1024 + 256);
#pragma unroll
                    for (int j = 0; j < CPW; ++j)
                        bfr[(ks + 1) & 1][j] = tr_pair_r(sb + rb0[ks + 1] + coloff[j], sb + rb1[ks + 1] + coloff[j]);
                }
                __builtin_amdgcn_sched_barrier(0);               // (keep the reads in front of the MFMAs they overlap with)
#pragma unroll
                for (int j = 0; j < CPW; ++j)                    // (a wave with one column tile fewer multiplies tile 0 again and drops it:
                    //  the wave with the full count sets the pace anyway, and the loop stays branch-free)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1], bfr[ks & 1][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++slot == SLOTS) slot = 0;
    }
    wait_vmcnt<0>();                                             // (the trailing fills are all out of range; nothing is left in flight at exit)
    if (a.ts) t_loop = __builtin_amdgcn_s_memrealtime();
    auto stamp = [&]() {
        if (a.ts && tid == 0) {
            unsigned long long* t = a.ts + (size_t)blockIdx.x * 8;
            t[0] = t_start; t[1] = t_setup; t[2] = t_first; t[3] = t_wait; t[4] = t_loop; t[5] = __builtin_amdgcn_s_memrealtime();
            t[6] = (unsigned long long)(u1 - u0) | (t_issue << 16); t[7] = (unsigned long long)CLS;
        }
    };

    // ---- accumulators -> this group's copy [N][Kc]: lane = channel of the tile, register e = small-side channel
    if (a.dbg & 1) { stamp(); return; }
    constexpr int Kc = G::NTAPS(CLS) * G::C;
    if (a.atomic_dst) {
        float* const dsta = a.atomic_dst + (size_t)(n0 + wr * 32 + 4 * h) * a.atomic_kpad + (lane & 31);
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int col = wc + j * WC;
            if (col < NCT) {
                const int k = col / CT, ct = col - k * CT;
                const int tap = (G::tap_ty(CLS, 0) + (k / NTX) * G::ST) * G::KW + G::tap_tx(CLS, 0) + (k % NTX) * G::ST;
                float* d = dsta + tap * G::C + ct * 32;
#pragma unroll
                for (int e = 0; e < 16; ++e) atomicAdd(d + (size_t)((e & 3) + 8 * (e >> 2)) * a.atomic_kpad, acc[j][e]);
            }
        }
        if (a.ts) { wait_vmcnt<0>(); stamp(); }
        return;
    }
    float* const dst = copy + (size_t)(n0 + wr * 32 + 4 * h) * Kc + (lane & 31);
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int col = wc + j * WC;
        if (col < NCT) {
            float* d = dst + col * 32;                           // column tile (k, ct) starts at k*C + ct*32 = col*32
#pragma unroll
            for (int e = 0; e < 16; ++e) d[(size_t)((e & 3) + 8 * (e >> 2)) * Kc] = acc[j][e];
        }
    }
    if (a.ts) { wait_vmcnt<0>(); stamp(); }
}


// Pair form (Geo::PAIR): 2*WAVES waves, the two classes of pair P behind ONE fill of the small image.  Waves 0..WAVES-1 multiply
// class A = pair_a(P), the others class B; the batch loop, its barrier and the DMA issue are common.
template <class G, int P>
__device__ __forceinline__ void wr_body_pair(const WrArgs& a, const int ms, const int u0, const int u1, const int g, char* const smem) {
    constexpr int WAVES = G::WAVES, W2 = 2 * WAVES, NF = G::P_NF, KST = G::KST, CT = G::CT, SLOT = G::P_SLOT, NTN = G::NTN, WC = G::WC;
    constexpr int CA = G::pair_a(P), CB = G::pair_b(P), NPIECES = G::p_bytes(P) / 1024, SMP = G::SM_USED / 1024;
    static_assert(G::SLOTS == 2, "the pair form waits for the whole fill at the barrier");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave / WAVES, w4 = wave - half * WAVES;
    unsigned long long t_start = 0, t_setup = 0, t_first = 0, t_wait = 0, t_loop = 0, t_issue = 0;
    if (a.ts) t_start = __builtin_amdgcn_s_memrealtime();

    int voff[NF];                                                // piece j of this wave = piece j*W2 + wave of the slot
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const wrgeo::Src s = G::src_pair(P, ((j * W2 + wave) * 64) + lane);
        voff[j] = s.tensor < 0 ? OOB : s.off;
    }
    const int n0 = ms * G::NS;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.S + n0), 0, (int)((size_t)a.nimg * G::OYX * G::N * 2 - (size_t)n0 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.Bg), 0, (int)((size_t)a.nimg * G::AH * G::AW * G::C * 2), 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem;
    auto issue = [&](int u, int slot) {
        const bool live = u < u1 && !(a.dbg & 4);
        const int soff_s = u * (G::IB * G::OYX * G::N * 2), soff_b = u * (G::IB * G::AH * G::AW * G::C * 2);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int piece = j * W2 + wave;                     // (wave-uniform)
            if (piece >= NPIECES) continue;
            const bool sm = piece < SMP;
            const __amdgpu_buffer_rsrc_t r = sm ? rs : rb;
            dma_1k(r, lds0 + slot * SLOT + piece * 1024, live ? voff[j] : OOB, sm ? soff_s : soff_b);
        }
    };
    issue(u0, 0);

    const int h = lane >> 5, q4 = (lane & 15) >> 2;
    const int lanecol = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int wr = w4 % NTN, wc = w4 / NTN;
    const int a_base = G::sm_off(wr, 8 * h + q4) + lanecol;
    auto stamp = [&](int cls) {
        if (a.ts && (tid & (WAVES * 64 - 1)) == 0 && half == 0) {
            unsigned long long* t = a.ts + (size_t)blockIdx.x * 8;
            t[0] = t_start; t[1] = t_setup; t[2] = t_first; t[3] = t_wait; t[4] = t_loop; t[5] = __builtin_amdgcn_s_memrealtime();
            t[6] = (unsigned long long)(u1 - u0) | (t_issue << 16); t[7] = (unsigned long long)cls;
        }
    };
    // one class's share of the workgroup: everything that depends on the class is inside; both shares run the same barriers
    auto run = [&](auto cls_tag, const int bg_off) {
        constexpr int CLS = decltype(cls_tag)::value;
        constexpr int CPW = G::CPW(CLS), NCT = G::NCT(CLS), NTX = G::NTX(CLS), LC = G::LC(CLS), Kc = G::NTAPS(CLS) * G::C;
        int rb0[KST], rb1[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const int kr = ks * 16 + 8 * h + q4;
            rb0[ks] = bg_off + G::rowcell(CLS, kr) * 64 + lanecol;
            rb1[ks] = bg_off + G::rowcell(CLS, kr + 4) * 64 + lanecol;
        }
        int coloff[CPW];
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int col = wc + j * WC, cc = col < NCT ? col : 0;
            const int k = cc / CT, ct = cc - k * CT;
            coloff[j] = (ct * G::NCELLP(CLS) + (k / NTX) * LC + k % NTX) * 64;
        }
        f32x16 acc[CPW];
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        int slot = 0;
        if (a.ts) t_setup = __builtin_amdgcn_s_memrealtime();
        for (int u = u0; u < u1; ++u) {
            unsigned long long tw = 0;
            if (a.ts) tw = __builtin_amdgcn_s_memrealtime();
            wait_vmcnt<0>();                                     // this wave's pieces of batch u have landed
            __builtin_amdgcn_s_barrier();                        // ... everybody's; and everybody is done reading batch u-1
            if (a.ts) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); if (u == u0) t_first = t; else t_wait += t - tw; }
            unsigned long long ti = 0;
            if (a.ts) ti = __builtin_amdgcn_s_memrealtime();
            issue(u + 1, slot ^ 1);                              // into the slot batch u-1 was read from
            if (a.ts) t_issue += __builtin_amdgcn_s_memrealtime() - ti;
            const char* const sb = smem + slot * SLOT;
            if (!(a.dbg & 2)) {
                bf16x8 af[2], bfr[2][CPW];
                af[0] = tr_pair_r(sb + a_base, sb + a_base + 256);
#pragma unroll
                for (int j = 0; j < CPW; ++j) bfr[0][j] = tr_pair_r(sb + rb0[0] + coloff[j], sb + rb1[0] + coloff[j]);
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    if (ks + 1 < KST) {
                        af[(ks + 1) & 1] = tr_pair_r(sb + a_base + (ks + 1) * 1024, sb + a_base + (ks + 1) * 1024 + 256);
#pragma unroll
                        for (int j = 0; j < CPW; ++j)
                            bfr[(ks + 1) & 1][j] = tr_pair_r(sb + rb0[ks + 1] + coloff[j], sb + rb1[ks + 1] + coloff[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < CPW; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1], bfr[ks & 1][j], acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            slot ^= 1;
        }
        wait_vmcnt<0>();
        if (a.ts) t_loop = __builtin_amdgcn_s_memrealtime();
        if (a.dbg & 1) { stamp(CLS); return; }
        if (a.atomic_dst) {
            float* const dsta = a.atomic_dst + (size_t)(n0 + wr * 32 + 4 * h) * a.atomic_kpad + (lane & 31);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int col = wc + j * WC;
                if (col < NCT) {
                    const int k = col / CT, ct = col - k * CT;
                    const int tap = (G::tap_ty(CLS, 0) + (k / NTX) * G::ST) * G::KW + G::tap_tx(CLS, 0) + (k % NTX) * G::ST;
                    float* d = dsta + tap * G::C + ct * 32;
#pragma unroll
                    for (int e = 0; e < 16; ++e) atomicAdd(d + (size_t)((e & 3) + 8 * (e >> 2)) * a.atomic_kpad, acc[j][e]);
                }
            }
        } else {
            float* const dst = a.slab + a.slab_off[CLS] + (size_t)g * G::N * Kc + (size_t)(n0 + wr * 32 + 4 * h) * Kc + (lane & 31);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int col = wc + j * WC;
                if (col < NCT) {
                    float* d = dst + col * 32;
#pragma unroll
                    for (int e = 0; e < 16; ++e) d[(size_t)((e & 3) + 8 * (e >> 2)) * Kc] = acc[j][e];
                }
            }
        }
        if (a.ts) { wait_vmcnt<0>(); stamp(CLS); }
    };
    if (half == 0) run(std::integral_constant<int, CA>{}, G::p_bga(P));
    else run(std::integral_constant<int, CB>{}, G::p_bgb(P));
}

template <int I, int N, typename F>
__device__ __forceinline__ void wr_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wr_static_for<I + 1, N>(f);
    }
}

template <class G>
__global__ __launch_bounds__(G::WAVES * 64 * (G::PAIR ? 2 : 1)) void wgrad_ring_kernel(const WrArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int jb = a.job[blockIdx.x];
    const int cls = jb >> 12, rel = jb & 0xfff;          // (pair form: `cls` is the pair, groups[] / job[] are per pair)
    const int ms = rel % G::MS, g = rel / G::MS;
    const int ng = a.groups[cls];
    const int u0 = (int)((long long)a.units * g / ng), u1 = (int)((long long)a.units * (g + 1) / ng);
    if constexpr (G::PAIR != 0) {
        wr_static_for<0, G::NPAIR>([&](auto ip) {
            constexpr int P = decltype(ip)::value;
            if (cls == P) wr_body_pair<G, P>(a, ms, u0, u1, g, smem);
        });
        return;
    }
    wr_static_for<0, G::NCLS>([&](auto ic) {
        constexpr int CLS = decltype(ic)::value;
        if (cls == CLS) {
            constexpr int Kc = G::NTAPS(CLS) * G::C;
            wr_body<G, CLS>(a, ms, u0, u1, a.slab + a.slab_off[CLS] + (size_t)g * G::N * Kc, smem);
        }
    });
}

// dst[n][tap*C + c] += sum over the class's copies of copy[n][k*C + c].  A block owns 32 float4 outputs; its 8 thread rows sum
// every 8th copy each (independent loads, a short chain: a copy loop per output was a latency chain of copies/8 round trips on
// a few dozen blocks) and meet in LDS.
constexpr int WRR_MAX = 24;
struct WrReduceArgs {
    struct Job { float* dst; const float* slab; int N, Kpad, C, Kc, copies, ntx, ty0, tx0, st, kw, first_block; } job[WRR_MAX];
    int n;
};
__global__ __launch_bounds__(256) void wgrad_ring_reduce_kernel(const WrReduceArgs a) {
    __shared__ f32x4 part[8][32];
    int j = 0;
    while (j + 1 < a.n && (int)blockIdx.x >= a.job[j + 1].first_block) ++j;
    const WrReduceArgs::Job& q = a.job[j];
    const int kv = q.Kc / 4;
    const int o = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const int t = (int)(blockIdx.x - q.first_block) * 32 + o;
    const bool in = t < q.N * kv;
    const int tt = in ? t : 0;
    const int n = tt / kv, k4 = (tt - n * kv) * 4;
    const float* src = q.slab + (size_t)n * q.Kc + k4;
    const size_t stride = (size_t)q.N * q.Kc;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int i = cl;
    for (; i + 24 < q.copies; i += 32) {                         // 4 independent 16-byte loads in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const f32x4*>(src + (size_t)(i + 8 * e) * stride);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; i < q.copies; i += 8) s += *reinterpret_cast<const f32x4*>(src + (size_t)i * stride);
    part[cl][o] = s;
    __syncthreads();
    if (cl != 0 || !in) return;
#pragma unroll
    for (int e = 1; e < 8; ++e) s += part[e][o];
    const int k = k4 / q.C, c = k4 - k * q.C;
    const int tap = (q.ty0 + (k / q.ntx) * q.st) * q.kw + q.tx0 + (k % q.ntx) * q.st;
    float* d = q.dst + (size_t)n * q.Kpad + tap * q.C + c;
    f32x4 ov = *reinterpret_cast<f32x4*>(d);
    ov += s;
    *reinterpret_cast<f32x4*>(d) = ov;
}

// the forward-form problem a geometry was compiled for
template <class G>
bool wr_matches(const WgradParams& p) {
    const GatherCommon& c = p.c;
    const GatherClass& k = p.cls[0];
    return c.nclasses == 1 && c.C == G::C && c.N == G::N && c.AH == G::AH && c.AW == G::AW && c.Ald == G::C && p.ldp == G::N &&
           k.OY == G::OH && k.OX == G::OW && c.OH == G::OH && c.OW == G::OW && k.TH == G::KH && k.TW == G::KW &&
           c.sy == G::ST && c.sx == G::ST && c.dy == 1 && c.dx == 1 && c.osy == 1 && c.osx == 1 && k.offy == -G::PAD && k.offx == -G::PAD &&
           k.ooy == 0 && k.oox == 0 && k.K == G::KH * G::KW * G::C;
}

template <class G>
int try_wr(const WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx) {
    if (!wr_matches<G>(p)) return 0;
    // pair form: opt-in.  Measured (gpurun_out/s2, DESIGN tried-and-lost): the loop is faster (hallucinate.6 without its epilogue 16.6
    // against 23.9 us: 65 instead of 45 MFMAs per SIMD and batch in the same 1.6 us), but at equal workgroup count every class has
    // twice the partial copies -- the atomic epilogue takes 6-9 us per workgroup instead of 3-5 -- and the step is 591-612 against 589 us
    if (G::PAIR && !mmvae_knob("wr_pair", 0)) return 0;          // (the one-class-per-workgroup geometry of the same layer follows in the list)
    if (mmvae_knob("dbg_skip_wgrad", 0) == 2) return 1;         // measurement aid: the step without the ring-staged weight gradients
    const GatherCommon& c = p.c;
    const int nimg = c.groups * c.group_n;
    if (nimg % G::IB != 0 || !p.cls[0].dWp || !ctx || !ctx->pool) return 0;
    if ((size_t)nimg * G::AH * G::AW * G::C * 2 >= (1ull << 31) || (size_t)nimg * G::OYX * G::N * 2 >= (1ull << 31)) return 0;
    const int units = nimg / G::IB;
    // image groups per class in proportion to the class's column tiles: every workgroup carries about the same number of MFMAs
    // (the 5x5 layer's classes hold 9 / 6 / 6 / 4 taps); the workgroup count of the launch: `target` below
    // in quarters of the CU count.  Inside the step a launch that leaves half the chip to the main chain costs the step less than one
    // that takes every CU for a shorter time (MultiMNIST: 652 -> 648 us per step at half the chip, 634 with ONE weight-gradient
    // stream); the 9-17 GFLOP layers of CelebA run shorter on the whole chip (1.873 against 1.888 ms per step)
    const double macs = (double)nimg * G::OYX * G::KH * G::KW * G::C * G::N;
    const int wq = mmvae_knob("wr_wgs", -1);            // (-1: not set.  A call site caches the value it looked up: the default must not vary)
    // (re-tuned after the critical-path work of the round's second half -- the second-modality stream is no longer waited for and the
    //  decoders' optimizer part runs in this stream's gap: three quarters of the chip 574.8 us per step, half 579.3, all of it 579.8)
    const int target = (G::WGQ > 0 ? mmvae_knob("wr_wgs_big", G::WGQ) : wq >= 0 ? wq : macs >= 4e9 ? 4 : 3) * mmvae_cu_count() / 4;
    // (a batch costs a workgroup a fixed part -- the DMA issue -- next to its MFMAs: measured 0.66 us + 0.10 us per column tile
    //  on hallucinate.6; knob wr_bias = the fixed part in column tiles)
    const int bias = mmvae_knob("wr_bias", 6);
    constexpr int NJ = G::PAIR ? G::NPAIR : G::NCLS;             // job classes: parity classes, or pairs of them
    auto wgt = [&](int i) {
        if (G::PAIR) return G::CPW(G::pair_a(i)) * G::WC + G::CPW(G::pair_b(i)) * G::WC + bias;
        return G::NCT(i) > 0 ? G::CPW(i) * G::WC + bias : 0;
    };
    int wsum = 0;
    for (int i = 0; i < NJ; ++i) wsum += wgt(i);
    WrArgs a{};
    a.S = p.P; a.Bg = c.A; a.units = units; a.nimg = nimg; a.dbg = mmvae_knob("wr_dbg", 0);
    a.ts = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)mmvae_knob("wr_ts_hi", 0) << 32) | (unsigned)mmvae_knob("wr_ts_lo", 0));
    size_t need = 0;
    int total = 0;
    int copies[WR_MAXCLS] = {};
    for (int i = 0; i < NJ; ++i) {
        int g = wgt(i) > 0 ? (int)((long long)target * wgt(i) / ((long long)wsum * G::MS)) : 0;
        g = wgt(i) > 0 ? std::max(1, std::min(g, units)) : 0;
        a.groups[i] = g; total += g * G::MS;
        if (G::PAIR) { copies[G::pair_a(i)] = g; copies[G::pair_b(i)] = g; } else copies[i] = g;
    }
    for (int i = 0; i < G::NCLS; ++i) {
        a.slab_off[i] = (long long)need;
        need += (size_t)copies[i] * G::N * G::NTAPS(i) * G::C;
    }
    if (total > WR_MAXJOBS) return 0;
    {   // Workgroup order.  The workgroups that read the same images (the classes and channel slices of one image range) should
        // share an XCD and run at the same time: the small image then crosses the fabric once and the other classes may hit it in
        // the XCD's L2.  Blocks b and b + 8 share an XCD (round-robin dispatch; speed only, never correctness): sort the jobs by
        // their first image and deal runs of RUN consecutive jobs to the 8 block residues in turn.
        struct J { int u0, code; };
        std::vector<J> js;
        for (int i = 0; i < NJ; ++i)
            for (int g = 0; g < a.groups[i]; ++g)
                for (int m = 0; m < G::MS; ++m) js.push_back(J{(int)((long long)units * g / a.groups[i]), (i << 12) | (g * G::MS + m)});
        std::stable_sort(js.begin(), js.end(), [](const J& x, const J& y) { return x.u0 < y.u0; });
        const int run = std::max(1, mmvae_knob("wr_run", NJ * G::MS));
        const int blk = 8 * run, full = total / blk * blk;
        for (int i = 0; i < total; ++i) {
            int b = i;
            if (i < full && mmvae_knob("wr_xcd", 1)) b = i / blk * blk + (i % run) * 8 + (i / run) % 8;
            a.job[b] = (unsigned short)js[i].code;
        }
    }
    // small dW (features.2: 128 KB, hallucinate.6: 200 KB): the groups' accumulators go into the packed gradient by fp32 atomics
    // (a few MB of adds in all, spread over the kernel's tail) -- no partial copies, no reduce launch behind the kernel.  Not
    // bit-reproducible from run to run (the order of the adds), like every atomically accumulated bias / BatchNorm gradient here.
    const bool atomic = (size_t)G::N * G::KH * G::KW * G::C * 4 <= (size_t)mmvae_knob("wr_atomic_kb", 256) * 1024;
    float* slab = atomic ? p.cls[0].dWp : ctx->take(need);
    if (!slab) return 0;
    a.slab = slab;
    if (atomic) { a.atomic_dst = p.cls[0].dWp; a.atomic_kpad = p.cls[0].Kpad; }
    for (int i = 0; i < G::NCLS && !atomic; ++i) {
        if (copies[i] == 0) continue;
        WgradRingJob j{};
        j.dst = p.cls[0].dWp; j.slab = slab + a.slab_off[i]; j.N = G::N; j.Kpad = p.cls[0].Kpad; j.C = G::C; j.Kc = G::NTAPS(i) * G::C;
        j.copies = copies[i]; j.ntx = G::NTX(i); j.ty0 = G::tap_ty(i, 0); j.tx0 = G::tap_tx(i, 0); j.st = G::ST; j.kw = G::KW;
        j.stream = stream;
        ctx->ring_jobs.push_back(j);
    }
    constexpr int lds = G::PAIR ? G::P_TOTAL : G::TOTAL;
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    MMVAE_LAUNCH((wgrad_ring_kernel<G>), dim3(total), dim3(G::WAVES * 64 * (G::PAIR ? 2 : 1)), (size_t)lds, stream, a);
    const int rc = mmvae_check_launch("wgrad_ring");
    return rc == MMVAE_OK ? 1 : rc;
}

}  // namespace

int try_launch_wgrad_ring(const WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx) {
    if (!mmvae_knob("wgrad_ring", 1)) return 0;
    if (p.c.a_bcast_n > 0 || p.c.a_mask || p.c.a_affine || p.c.a_act != ACT_NONE || p.p_affine || p.p_act != ACT_NONE) return 0;
    int rc;
#define X(name, ...) if ((rc = try_wr<wrgeo::Geo<__VA_ARGS__>>(p, stream, ctx)) != 0) return rc;
    WGRAD_RING_GEOS(X)
#undef X
    return 0;
}

int launch_wgrad_ring_reduce(WgradSlabCtx* ctx, hipStream_t stream, bool only_own) {
    if (!ctx || ctx->ring_jobs.empty()) return MMVAE_OK;
    std::vector<WgradRingJob> todo, keep;
    for (const WgradRingJob& j : ctx->ring_jobs) (only_own && j.stream != stream ? keep : todo).push_back(j);
    ctx->ring_jobs.swap(keep);
    size_t done = 0;
    while (done < todo.size()) {
        WrReduceArgs a{};
        int blocks = 0;
        while (done < todo.size() && a.n < WRR_MAX) {
            const WgradRingJob& j = todo[done++];
            WrReduceArgs::Job& q = a.job[a.n++];
            q.dst = j.dst; q.slab = j.slab; q.N = j.N; q.Kpad = j.Kpad; q.C = j.C; q.Kc = j.Kc; q.copies = j.copies; q.ntx = j.ntx;
            q.ty0 = j.ty0; q.tx0 = j.tx0; q.st = j.st; q.kw = j.kw; q.first_block = blocks;
            blocks += (j.N * (j.Kc / 4) + 31) / 32;
        }
        MMVAE_LAUNCH(wgrad_ring_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
        MMVAE_TRY(mmvae_check_launch("wgrad_ring_reduce"));
    }
    return MMVAE_OK;
}
