// Image-resident weight-gradient kernel for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   dW[n][(tap, c)] = sum over pixel rows of  S[row][n] * G[row gathered at tap][c]
// in the forward-form geometry of convres_geo.h (FORM 0): rows are the pixels of the small side of a layer, S is the tensor
// living there (Conv2d: the gradient of the layer's raw output; ConvTranspose2d: the layer's activated input), G the tensor of
// the big side (the activated input / the output gradient).
//
// wgrad_kernel (gemm.hip) streams 64 pixel rows per iteration through LDS: every tap class re-reads the gathered rows from L2
// (4.3x the algorithmic traffic on hallucinate.6), each staged vector carries its own address arithmetic, and the partial
// tiles of ~1000 workgroups meet in a reduce launch.  Here a workgroup keeps BOTH tensors of NI images resident in LDS:
//   * every byte crosses L2 -> CU once per workgroup column (N is split over gridDim.y when the accumulators would not fit);
//   * both MFMA operands need the pixel row on the k axis: ds_read_b64_tr_b16 reads them transposed straight out of the
//     NHWC images; the gathered operand's tap is an immediate on a per-lane row base (same LDS layout as the forward
//     kernels, x de-interleaved into stride-parity planes);
//   * the whole dW of the workgroup's column lives in accumulators ([n-tile][tap, channel tile] per wave) across a
//     PERSISTENT loop over image sets: one slab copy per workgroup (64-128 per layer instead of ~1000), summed by the
//     existing wgrad_reduce_kernel;
//   * the operands are transformed while they are staged (GatherTransform): Swish(BatchNorm(raw)) of an activation and the
//     BatchNorm-backward of a gradient are never materialised -- bn_act / bn_bwd_apply leave the step entirely for these
//     layers (the BatchNorm parameter gradients are added here by workgroup (0, 0)).
#include "gemm.h"
#include "convres_geo.h"
#include "convres.h"
#include "bn_dev.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4c;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4c;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ float swish_fast_w(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

struct StageTr {                // staging transform of one operand (GatherTransform, device side)
    BnFinalizeArgs fin;         // kind 1
    const bf16* r;              // kind 2
    const float2* red; const float2* mr; const float* gamma;
    float* dgamma; float* dbeta;
    float inv_cnt; int groups;
};

struct CrWgradArgs {
    const bf16* S;              // small-side tensor [nimg][OH*OW][N]
    const bf16* Bg;             // big-side tensor [nimg][AH][AW][C]
    int nimg, group_n;
    float* slab;                // [gridDim.x][N][Kpad] partial gradients (plain stores)
    int Kpad;
    StageTr ts, tb;
};

// two transposed 8-byte reads = one 32x32x16 operand fragment whose k axis runs over LDS rows
__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)a0);
    u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4c*)a1);
    return u.v;
}

// TRS / TRB: staging transform of the small / big tensor (0 none, 1 BatchNorm + Swish of a raw tensor, 2 BatchNorm backward)
// MAXG: BatchNorm groups (passes) the coefficient tables are sized for
constexpr int CRW_MAXG = 3;

template <class G, int NI, int WAVES>
struct CrwLayout {
    static constexpr int N = G::N, C = G::C;
    static constexpr int OYX = G::OYX(0), ROWS = NI * OYX, KST = crgeo::cdiv(ROWS, 16);
    static constexpr int SP = N * 2 + 16;                    // small image: bytes per pixel row
    static constexpr int SIMG = KST * 16 * SP;               // NI small images, rows padded to whole k-steps (zero rows)
    static constexpr int BIMG = NI * G::IMG_BYTES;
    static constexpr int BUF = BIMG + SIMG;                  // one image-set buffer; two of them (double buffering)
    static constexpr int TRT = CRW_MAXG * (N + C) * 16;      // staging-transform coefficients of every group
    static constexpr int TOTAL = 2 * BUF + TRT;
    static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

template <class G, int NI, int WAVES, int NSPLIT, int TRS, int TRB>
__global__ __launch_bounds__(WAVES * 64) void convres_wgrad_kernel(const CrWgradArgs a) {
    static_assert(G::FORM == 0, "weight gradients use the forward-form geometry");
    using L = CrwLayout<G, NI, WAVES>;
    constexpr int NTHR = WAVES * 64, N = G::N, C = G::C;
    constexpr int OYX = L::OYX, ROWS = L::ROWS, KST = L::KST, SP = L::SP, BIMG = L::BIMG, BUF = L::BUF;
    constexpr int NTL = N / 32 / NSPLIT;                     // n-tiles of this workgroup
    constexpr int CPT = C / 32, TAPS = G::TH(0) * G::TW(0), CTL = TAPS * CPT;           // column tiles (tap, channel block)
    constexpr int CTW = crgeo::cdiv(CTL, WAVES);             // column tiles per wave: w, w + WAVES, ...
    static_assert(N % (32 * NSPLIT) == 0 && C % 32 == 0, "tile counts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* const trt_s = reinterpret_cast<float4*>(smem + 2 * BUF);                     // [MAXG][N]
    float4* const trt_b = trt_s + CRW_MAXG * N;                                           // [MAXG][C]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int n0 = blockIdx.y * NTL * 32;
    const int nsets = a.nimg / NI;
    const int ngroups = a.nimg / a.group_n;

    // ---- per-lane row bases of the gathered operand: the lane ADDRESSES rows 16ks + 8h + q and + 4 of every k-step
    int gb0[KST], gb1[KST];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int R = ks * 16 + 8 * h + q + 4 * e;
            R = R < ROWS ? R : 0;                            // padding rows meet zero rows of the small image: any finite data
            const int img = R / OYX, qq = R - img * OYX;
            const int jy = qq / G::OX(0), jx = qq - jy * G::OX(0);
            const int v = img * G::IMG_BYTES + jy * G::row_stride(0) + jx * G::col_stride(0) + (16 * (g & 1) + 4 * p) * 2;
            if (e == 0) gb0[ks] = v; else gb1[ks] = v;
        }
    }
    const int sa0 = BIMG + (8 * h + q) * SP + (16 * (g & 1) + 4 * p) * 2 + n0 * 2;      // small image: same two rows, linear

    f32x16 acc[NTL][CTW];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- staging slots: vector v of a set's big tensor (slots [0, VB)) or small tensor (slots [VB, VT)) per thread
    constexpr int VPB = C / 8, VPS = N / 8;                  // 16-byte vectors per pixel
    constexpr int NVB = NI * G::AH * G::AW * VPB, NVS = ROWS * VPS;
    constexpr int VB = crgeo::cdiv(NVB, NTHR), VS = crgeo::cdiv(NVS, NTHR), VT = VB + VS;
    static_assert(NTHR % VPB == 0 && NTHR % VPS == 0, "a thread keeps its channel vector across staging slots");
    constexpr int B1 = crgeo::cdiv(VT, 3), B2 = crgeo::cmin(VT, 2 * B1);               // batches [0,B1) [B1,B2) [B2,VT)
    i32x4c xr[B1], yr[(TRS == 2 || TRB == 2) ? B1 : 1];
    auto slot_load = [&](auto ji, auto bi, int set) {        // global -> registers (register bi of the batch)
        constexpr int j = decltype(ji)::value, b = decltype(bi)::value;
        if constexpr (j < VB) {
            const int v = tid + j * NTHR;
            const size_t o = (size_t)set * NI * (G::AH * G::AW * C) + (size_t)v * 8;
            if (v < NVB) {
                xr[b] = *reinterpret_cast<const i32x4c*>(a.Bg + o);
                if constexpr (TRB == 2) yr[b] = *reinterpret_cast<const i32x4c*>(a.tb.r + o);
            }
        } else {
            const int v = tid + (j - VB) * NTHR;
            const size_t o = (size_t)set * NI * (OYX * N) + (size_t)v * 8;
            if (v < NVS) {
                xr[b] = *reinterpret_cast<const i32x4c*>(a.S + o);
                if constexpr (TRS == 2) yr[b] = *reinterpret_cast<const i32x4c*>(a.ts.r + o);
            }
        }
    };
    auto xform = [&](auto trc, i32x4c x, i32x4c y, const float4* cf) {
        constexpr int TR = decltype(trc)::value;
        if constexpr (TR == 0) return x;
        const bf16x8 xv = __builtin_bit_cast(bf16x8, x), yv = __builtin_bit_cast(bf16x8, y);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float4 t = cf[e];
            if constexpr (TR == 1) o[e] = (bf16)swish_fast_w((float)xv[e] * t.x + t.y);
            else o[e] = (bf16)((float)xv[e] * t.x + ((float)yv[e] * t.y + t.z));
        }
        return __builtin_bit_cast(i32x4c, o);
    };
    auto slot_store = [&](auto ji, auto bi, int set, char* buf) {      // registers -> transform -> LDS buffer
        constexpr int j = decltype(ji)::value, b = decltype(bi)::value;
        const int grp = (set * NI) / a.group_n;
        if constexpr (j < VB) {
            const int v = tid + j * NTHR;
            if (v < NVB) {
                const int pix = v / VPB, cv = v - pix * VPB;
                const int img = pix / (G::AH * G::AW), p2 = pix - img * (G::AH * G::AW);
                const int iy = p2 / G::AW, ix = p2 - iy * G::AW;
                const int cell = G::cell(iy, ix);
                const i32x4c val = xform(std::integral_constant<int, TRB>{}, xr[b], yr[TRB == 2 ? b : 0], trt_b + grp * C + cv * 8);
                if (cell >= 0) *reinterpret_cast<i32x4c*>(buf + img * G::IMG_BYTES + cell + cv * 16) = val;
            }
        } else {
            const int v = tid + (j - VB) * NTHR;
            if (v < NVS) {
                const int row = v / VPS, cv = v - row * VPS;
                const i32x4c val = xform(std::integral_constant<int, TRS>{}, xr[b], yr[TRS == 2 ? b : 0], trt_s + grp * N + cv * 8);
                *reinterpret_cast<i32x4c*>(buf + BIMG + row * SP + cv * 16) = val;
            }
        }
    };
    // batch k of set `set`: loads / stores of its slots
    auto batch_load = [&](auto ki, int set) {
        constexpr int k = decltype(ki)::value, lo = k == 0 ? 0 : (k == 1 ? B1 : B2), hi = k == 0 ? B1 : (k == 1 ? B2 : VT);
        static_for<lo, hi>([&](auto ji) { slot_load(ji, std::integral_constant<int, decltype(ji)::value - lo>{}, set); });
    };
    auto batch_store = [&](auto ki, int set, char* buf) {
        constexpr int k = decltype(ki)::value, lo = k == 0 ? 0 : (k == 1 ? B1 : B2), hi = k == 0 ? B1 : (k == 1 ? B2 : VT);
        static_for<lo, hi>([&](auto ji) { slot_store(ji, std::integral_constant<int, decltype(ji)::value - lo>{}, set, buf); });
    };
    // k-steps [K0, K1) of the set in `buf`: dW += S^T G
    auto compute = [&](auto k0i, auto k1i, const char* buf) {
        constexpr int K0 = decltype(k0i)::value, K1 = decltype(k1i)::value;
        static_for<K0, K1>([&](auto ksi) {
            constexpr int ks = decltype(ksi)::value;
            bf16x8 af[NTL];
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
                const char* s0 = buf + sa0 + ks * 16 * SP + i * 64;
                af[i] = tr_pair(s0, s0 + 4 * SP);
            }
#pragma unroll
            for (int j = 0; j < CTW; ++j) {
                const int ct = wave + j * WAVES;             // wave-uniform
                if (ct < CTL) {
                    const int tap = ct / CPT, cb = ct - tap * CPT;
                    const int ty = tap / G::TW(0), tx = tap - ty * G::TW(0);
                    const int toff = G::tap_off(0, ty, tx) + cb * 64;
                    const bf16x8 bf = tr_pair(buf + gb0[ks] + toff, buf + gb1[ks] + toff);
#pragma unroll
                    for (int i = 0; i < NTL; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf, acc[i][j], 0, 0, 0);
                }
            }
        });
    };

    // ---- prologue: first set's loads in flight while both buffers are zeroed (rings, padding rows) and the coefficient
    //      tables of every group are made
    int set = blockIdx.x;
    if (set < nsets) batch_load(std::integral_constant<int, 0>{}, set);
    {
        const i32x4c z = {0, 0, 0, 0};
        for (int i = tid * 16; i < 2 * BUF; i += NTHR * 16) *reinterpret_cast<i32x4c*>(smem + i) = z;
        auto coef = [&](auto trc, const StageTr& t, int nch, float4* dst, bool params) {
            constexpr int TR = decltype(trc)::value;
            if constexpr (TR == 1) {
                for (int i = tid; i < ngroups * nch; i += NTHR) {
                    float2 aff, mr;
                    bn_channel_tables(t.fin, i / nch, i % nch, aff, mr);
                    dst[i] = make_float4(aff.x, aff.y, 0.f, 0.f);
                }
            } else if constexpr (TR == 2) {
                for (int i = tid; i < ngroups * nch; i += NTHR) {
                    const int gi = i / nch, ch = i - gi * nch;
                    float sx = 0.f, sy = 0.f;
                    for (int sl = 0; sl < MMVAE_STAT_SLOTS; ++sl) {
                        const float2 v = t.red[((size_t)gi * MMVAE_STAT_SLOTS + sl) * nch + ch];
                        sx += v.x; sy += v.y;
                    }
                    const float2 mr = t.mr[i];
                    const float gg = t.gamma[ch] * mr.y, m1 = sx * t.inv_cnt, m2 = sy * t.inv_cnt;
                    const float cb = -gg * m2 * mr.y;
                    // dr = gg*(db - m1 - (r - mean)*rstd*m2) = gg*db + cb*r + (-cb*mean - gg*m1)
                    dst[i] = make_float4(gg, cb, -cb * mr.x - gg * m1, 0.f);
                }
                if (params && (t.dgamma || t.dbeta)) {       // BatchNorm parameter gradients: once per layer
                    for (int ch = tid; ch < nch; ch += NTHR) {
                        float tg = 0.f, tb2 = 0.f;
                        for (int gi = 0; gi < ngroups; ++gi)
                            for (int sl = 0; sl < MMVAE_STAT_SLOTS; ++sl) {
                                const float2 v = t.red[((size_t)gi * MMVAE_STAT_SLOTS + sl) * nch + ch];
                                tb2 += v.x; tg += v.y;
                            }
                        if (t.dgamma) t.dgamma[ch] += tg;
                        if (t.dbeta) t.dbeta[ch] += tb2;
                    }
                }
            }
        };
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;
        coef(std::integral_constant<int, TRS>{}, a.ts, N, trt_s, first);
        coef(std::integral_constant<int, TRB>{}, a.tb, C, trt_b, first);
    }
    __syncthreads();
    char* cur = smem;
    char* nxt = smem + BUF;
    if (set < nsets) {
        batch_store(std::integral_constant<int, 0>{}, set, cur);
        batch_load(std::integral_constant<int, 1>{}, set); batch_store(std::integral_constant<int, 1>{}, set, cur);
        batch_load(std::integral_constant<int, 2>{}, set); batch_store(std::integral_constant<int, 2>{}, set, cur);
    }
    __syncthreads();

    // ---- persistent loop: the next set is staged into the other buffer in three batches between thirds of the k-loop
    constexpr int KA = KST / 3, KB = 2 * KST / 3;
    for (; set < nsets; set += gridDim.x) {
        const int ns = set + gridDim.x;
        const bool more = ns < nsets;
        if (more) batch_load(std::integral_constant<int, 0>{}, ns);
        compute(std::integral_constant<int, 0>{}, std::integral_constant<int, KA>{}, cur);
        if (more) { batch_store(std::integral_constant<int, 0>{}, ns, nxt); batch_load(std::integral_constant<int, 1>{}, ns); }
        compute(std::integral_constant<int, KA>{}, std::integral_constant<int, KB>{}, cur);
        if (more) { batch_store(std::integral_constant<int, 1>{}, ns, nxt); batch_load(std::integral_constant<int, 2>{}, ns); }
        compute(std::integral_constant<int, KB>{}, std::integral_constant<int, KST>{}, cur);
        if (more) batch_store(std::integral_constant<int, 2>{}, ns, nxt);
        __syncthreads();                                     // next set complete, this one consumed
        char* t = cur; cur = nxt; nxt = t;
    }

    // ---- partial gradient of this workgroup: slab[blockIdx.x][n][k], k = tap*C + channel (the packed weight layout)
    float* const dst = a.slab + (size_t)blockIdx.x * N * a.Kpad;
#pragma unroll
    for (int j = 0; j < CTW; ++j) {
        const int ct = wave + j * WAVES;
        if (ct < CTL) {
            const int k = ct * 32 + r;                       // (tap*CPT + cb)*32 + r = tap*C + cb*32 + r
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    dst[(size_t)n * a.Kpad + k] = acc[i][j][e];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <class G>
bool wgeo_matches(const WgradParams& p) {
    const GatherCommon& c = p.c;
    if (c.nclasses != 1 || c.C != G::C || c.Ald != G::C || c.N != G::N || p.ldp != G::N) return false;
    if (c.AH != G::AH || c.AW != G::AW || c.OH != G::OH || c.OW != G::OW) return false;
    if (c.sy != G::S || c.sx != G::S || c.dy != 1 || c.dx != 1 || c.osy != 1 || c.osx != 1) return false;
    const GatherClass& k = p.cls[0];
    return k.OY == G::OY(0) && k.OX == G::OX(0) && k.TH == G::TH(0) && k.TW == G::TW(0) && k.offy == G::offy(0) &&
           k.offx == G::offx(0) && k.ooy == 0 && k.oox == 0 && k.K == G::K(0) && k.Kpad >= G::K(0);
}

int fill_tr(const GatherTransform* t, StageTr& o, int nch, int groups, const char* which) {
    if (!t || t->kind == 0) return 0;
    if (t->kind == 1) {
        o.fin = t->fin;
        o.fin.running_mean = nullptr; o.fin.running_var = nullptr; o.fin.num_batches_tracked = nullptr;   // already updated by the forward
        MMVAE_REQUIRE(o.fin.C == nch && o.fin.G == groups && o.fin.gamma && o.fin.beta, "convres wgrad: BatchNorm tables of the %s operand", which);
        return 1;
    }
    MMVAE_REQUIRE(t->r && t->red && t->mr && t->gamma && t->groups == groups, "convres wgrad: BatchNorm-backward staging of the %s operand", which);
    o.r = t->r; o.red = t->red; o.mr = t->mr; o.gamma = t->gamma; o.dgamma = t->dgamma; o.dbeta = t->dbeta;
    o.inv_cnt = t->inv_cnt; o.groups = t->groups;
    return 2;
}

template <class G, int NI, int WAVES, int NSPLIT, int TRS, int TRB>
int launch_crw(const CrWgradArgs& a, int chunks, hipStream_t stream) {
    constexpr size_t lds = CrwLayout<G, NI, WAVES>::TOTAL;
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&convres_wgrad_kernel<G, NI, WAVES, NSPLIT, TRS, TRB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((convres_wgrad_kernel<G, NI, WAVES, NSPLIT, TRS, TRB>), dim3(chunks, NSPLIT), dim3(WAVES * 64), lds, stream, a);
    return mmvae_check_launch("convres_wgrad");
}

// TRS/TRB combinations the step uses: (0,0) materialised operands; Conv2d: small = gradient (2), big = activation (0 or 1);
// ConvTranspose2d (forward-form weight gradient): small = activation (1), big = gradient (2)
template <class G, int NI, int WAVES, int NSPLIT>
int try_crw(WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx) {
    if (!wgeo_matches<G>(p)) return 0;
    const GatherCommon& c = p.c;
    const int nimg = c.groups * c.group_n;
    if (c.group_n % NI != 0 || !ctx || !ctx->pool || c.groups > CRW_MAXG) return 0;
    CrWgradArgs a{};
    const int ks = fill_tr(p.trP, a.ts, G::N, c.groups, "plain");
    const int kb = fill_tr(p.trA, a.tb, G::C, c.groups, "gathered");
    if (ks < 0 || kb < 0) return ks < 0 ? ks : kb;
    const int nsets = nimg / NI;
    int chunks = nsets < 96 ? nsets : 96;
    const int Kpad = p.cls[0].Kpad;
    const size_t slab_elems = (size_t)G::N * Kpad;
    while (chunks > 1 && ctx->used + (size_t)chunks * slab_elems > ctx->cap) chunks /= 2;
    float* slab = ctx->take((size_t)chunks * slab_elems);
    if (!slab) return 0;
    a.S = p.P; a.Bg = c.A; a.nimg = nimg; a.group_n = c.group_n; a.slab = slab; a.Kpad = Kpad;
    WgradSlabJob j{};
    j.dst = p.cls[0].dWp; j.slab = slab; j.N = G::N; j.K = G::K(0); j.Kpad = Kpad; j.chunks = chunks; j.chunk_stride = (long long)slab_elems;
    j.stream = stream;
    ctx->jobs.push_back(j);
    int rc;
    if (ks == 0 && kb == 0) rc = launch_crw<G, NI, WAVES, NSPLIT, 0, 0>(a, chunks, stream);
    else if (ks == 2 && kb == 0) rc = launch_crw<G, NI, WAVES, NSPLIT, 2, 0>(a, chunks, stream);
    else if (ks == 2 && kb == 1) rc = launch_crw<G, NI, WAVES, NSPLIT, 2, 1>(a, chunks, stream);
    else if (ks == 1 && kb == 2) rc = launch_crw<G, NI, WAVES, NSPLIT, 1, 2>(a, chunks, stream);
    else { mmvae_set_error("convres wgrad: staging-transform combination (%d, %d) is not compiled", ks, kb); return MMVAE_EINVAL; }
    return rc == MMVAE_OK ? 1 : rc;
}

// =====================================================================================================================
// Tap-group weight gradient: many SMALL workgroups, whole pixels, plain-store partial sums.
//
// dW is [N][taps][C].  A workgroup owns one slice of it -- 32 output channels x TG taps x all C gathered channels -- and one
// group of images, which it streams through LDS one batch (IB images) at a time: the whole big-side image (same de-interleaved
// layout as the forward kernels, pixel pitch C*2+16) and 32 of the N channels of the small side.  The pixel rows of the batch
// run down the k axis of 32x32x16 MFMAs, both operands through transposed LDS reads (the gathered operand's tap is a per-lane
// address offset, a column tile is 32 channels of one tap).  At the end the 32 x TG x C accumulators go to the image group's
// slab copy with plain 128-byte row stores; the existing reduce kernel sums the copies (as many as there are image groups).
// What the two earlier designs taught: keep whole pixels (16-byte channel slices pulled 4-8x their bytes
// through L2), no float atomics (~180 G/s), and workgroups small enough (< 80 KB, 4 waves) to sit on a CU next to the main
// chain's kernels; the price is that an image is staged by (N/32) x (tap groups) workgroups.
struct WtArgs {
    const bf16* S;              // small-side tensor [nimg][OH*OW][N]
    const bf16* Bg;             // big-side tensor [nimg][AH][AW][C]
    float* slab;                // [image groups][N][Kpad] partial gradients, k = tap*C + c (plain stores)
    int Kpad;
    int ipg;                    // images per image group (a multiple of IB)
    int dbg;
};

template <class G, int IB>
struct WtLayout {
    static constexpr int OYX = G::OH * G::OW, ROWS = IB * OYX, KST = crgeo::cdiv(ROWS, 16), RPAD = KST * 16;
    static constexpr int DP = 80;                                        // small-side row pitch: 32 x bf16 + 16
    static constexpr int A_BYTES = (IB * G::IMG_BYTES + 15) / 16 * 16;
    static constexpr int OFF_A = 0, OFF_D = A_BYTES, OFF_T = OFF_D + RPAD * DP, TOTAL = OFF_T + RPAD * 4;
    static_assert(TOTAL <= 80 * 1024, "the tap-group weight gradient is meant to share a CU");
};

template <class G, int TG, int IB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void wgrad_tap_kernel(const WtArgs a) {
    using L = WtLayout<G, IB>;
    constexpr int NTHR = WAVES * 64, NMS = G::N / 32, TAPS = G::KH * G::KW, NTG = crgeo::cdiv(TAPS, TG), CT = G::C / 32;
    constexpr int NTL = TG * CT, NTW = crgeo::cdiv(NTL, WAVES), OYX = L::OYX, VPP = G::C / 8;
    static_assert(G::C % 32 == 0, "a column tile is 32 channels of one tap");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_s = smem + L::OFF_A;
    char* const d_s = smem + L::OFF_D;
    int* const tab = reinterpret_cast<int*>(smem + L::OFF_T);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = blockIdx.x % NTG, ms = (blockIdx.x / NTG) % NMS, gi = blockIdx.x / (NTG * NMS);
    const int img_begin = gi * a.ipg;
    const int tap0 = tg * TG, ntaps = TAPS - tap0 < TG ? TAPS - tap0 : TG;      // this workgroup's taps

    // ---- staging: fetch a batch into registers (in flight during the previous batch's MFMAs), store after the barrier
    constexpr int NVA = IB * G::AH * G::AW * VPP, ITA = crgeo::cdiv(NVA, NTHR);
    constexpr int NVD = L::ROWS * 4, ITD = crgeo::cdiv(NVD, NTHR);
    i32x4c ra[ITA], rd[ITD];
    auto fetch = [&](int img0) {
        const bf16* srcA = a.Bg + (size_t)img0 * (G::AH * G::AW * G::C);
        const bf16* srcD = a.S + (size_t)img0 * (OYX * G::N) + ms * 32;
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            const int v = tid + it * NTHR;
            if (v < NVA) ra[it] = *reinterpret_cast<const i32x4c*>(srcA + (size_t)v * 8);
        }
#pragma unroll
        for (int it = 0; it < ITD; ++it) {
            const int v = tid + it * NTHR;
            if (v < NVD) rd[it] = *reinterpret_cast<const i32x4c*>(srcD + (size_t)(v >> 2) * G::N + (v & 3) * 8);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            const int v = tid + it * NTHR;
            if (v < NVA) {
                const int pix = v / VPP, cv = v - pix * VPP;
                const int img = pix / (G::AH * G::AW), p2 = pix - img * (G::AH * G::AW);
                const int iy = p2 / G::AW, ix = p2 - iy * G::AW;
                const int cell = G::cell(iy, ix);
                if (cell >= 0) *reinterpret_cast<i32x4c*>(a_s + img * G::IMG_BYTES + cell + cv * 16) = ra[it];
            }
        }
#pragma unroll
        for (int it = 0; it < ITD; ++it) {
            const int v = tid + it * NTHR;
            if (v < NVD) *reinterpret_cast<i32x4c*>(d_s + (v >> 2) * L::DP + (v & 3) * 16) = rd[it];
        }
    };
    fetch(img_begin);
    {   // zero the image area and the padding rows of the small-side tile once (neither is written again); per-row bases of
        // the gathered operand (padding rows: row 0's -- their small-side rows are zero, the product vanishes)
        const i32x4c z = {0, 0, 0, 0};
        for (int i = tid * 16; i < L::A_BYTES; i += NTHR * 16) *reinterpret_cast<i32x4c*>(a_s + i) = z;
        for (int i = L::ROWS * L::DP + tid * 16; i < L::RPAD * L::DP; i += NTHR * 16) *reinterpret_cast<i32x4c*>(d_s + i) = z;
        for (int e = tid; e < L::RPAD; e += NTHR) {
            const int ee = e < L::ROWS ? e : 0;
            const int img = ee / OYX, q = ee - img * OYX, jy = q / G::OW, jx = q - jy * G::OW;
            tab[e] = img * G::IMG_BYTES + jy * G::row_stride(0) + jx * G::col_stride(0);
        }
    }
    // ---- per-lane constants of the transposed reads: lane (g4, q, p) of a 16-lane group addresses k-row q, columns 4p..4p+3
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5;
    const int col0 = 16 * (g4 & 1) + 4 * p4;                 // first of this lane's 4 channels inside a 32-channel tile
    int boff[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int t = wave + j * WAVES;                      // tile = (tap of the group, channel tile)
        const int tl = t / CT, ct = t - tl * CT;
        const int tap = tap0 + (tl < ntaps ? tl : 0);        // (tiles past the group's last tap: any valid address, results dropped)
        boff[j] = G::tap_off(0, tap / G::KW, tap % G::KW) + (ct * 32 + col0) * 2;
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int nb = a.ipg / IB;
    for (int b = 0; b < nb; ++b) {
        __syncthreads();                                     // the previous batch's readers are done (first pass: the zero fill)
        store();
        if (b + 1 < nb) fetch(img_begin + (b + 1) * IB);
        __syncthreads();
#pragma unroll 2
        for (int ks = (a.dbg & 2) ? L::KST : 0; ks < L::KST; ++ks) {
            const int row = ks * 16 + 8 * h + q4;
            const int ro0 = tab[row], ro1 = tab[row + 4];
            const char* s0 = d_s + row * L::DP + col0 * 2;
            const bf16x8 af = tr_pair(s0, s0 + 4 * L::DP);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (wave + j * WAVES < NTL) {                // (uniform per wave)
                    const bf16x8 bf = tr_pair(a_s + ro0 + boff[j], a_s + ro1 + boff[j]);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // ---- accumulators -> this image group's slab copy: lane = channel of the tile's tap, register e = output channel
    const int n = lane & 31;
    float* const dst = a.slab + (size_t)gi * G::N * a.Kpad + (size_t)(ms * 32) * a.Kpad;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int t = wave + j * WAVES, tl = t / CT, ct = t - tl * CT;
        if (t < NTL && tl < ntaps && !(a.dbg & 1)) {
            float* d = dst + (tap0 + tl) * G::C + ct * 32 + n;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[(size_t)((e & 3) + 8 * (e >> 2) + 4 * h) * a.Kpad] = acc[j][e];
        }
    }
}

template <class G, int TG, int IB, int WAVES>
int try_wt(const WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx) {
    if (!wgeo_matches<G>(p)) return 0;
    const GatherCommon& c = p.c;
    const int nimg = c.groups * c.group_n;
    if (nimg % IB != 0 || !p.cls[0].dWp || !ctx || !ctx->pool) return 0;
    constexpr int TAPS = G::KH * G::KW, slices = (G::N / 32) * crgeo::cdiv(TAPS, TG);
    const int nb = nimg / IB, Kpad = p.cls[0].Kpad;
    // image groups: about ws_wgs workgroups per CU in flight, every group a whole number of batches, slab copies within the pool
    int groups = mmvae_knob("ws_wgs", 2) * mmvae_cu_count() / slices;
    if (groups < 1) groups = 1;
    if (groups > nb) groups = nb;
    while (nb % groups != 0) --groups;
    const size_t slab_elems = (size_t)G::N * Kpad;
    while (groups > 1 && (ctx->used + (size_t)groups * slab_elems > ctx->cap || nb % groups != 0)) --groups;
    float* slab = ctx->take((size_t)groups * slab_elems);
    if (!slab) return 0;
    WtArgs a{};
    a.S = p.P; a.Bg = c.A; a.slab = slab; a.Kpad = Kpad; a.ipg = nimg / groups; a.dbg = mmvae_knob("ws_dbg", 0);
    WgradSlabJob j{};
    j.dst = p.cls[0].dWp; j.slab = slab; j.N = G::N; j.K = G::K(0); j.Kpad = Kpad; j.chunks = groups; j.chunk_stride = (long long)slab_elems;
    j.stream = stream;
    ctx->jobs.push_back(j);
    constexpr size_t lds = WtLayout<G, IB>::TOTAL;
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tap_kernel<G, TG, IB, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MMVAE_LAUNCH((wgrad_tap_kernel<G, TG, IB, WAVES>), dim3(groups * slices), dim3(WAVES * 64), lds, stream, a);
    const int rc = mmvae_check_launch("wgrad_tap");
    return rc == MMVAE_OK ? 1 : rc;
}

using crgeo::Geo;
//                FORM C   N    AH  AW  OH  OW KH KW S PAD PIXPAD
typedef Geo<0, 32, 64, 25, 25, 12, 12, 4, 4, 2, 1, 0> W_mm_conv2;      // MultiMNIST features.2
typedef Geo<0, 64, 128, 12, 12, 6, 6, 4, 4, 2, 1, 0> W_mm_conv3;       // features.5 and hallucinate.3 (forward-form geometry)
typedef Geo<0, 32, 64, 25, 25, 12, 12, 5, 5, 2, 1, 0> W_mm_convT3;     // hallucinate.6

}  // namespace

// forward-form geometries of the weight gradients with the forward kernels' pixel pitch (Conv2d: gathered = the layer input;
// ConvTranspose2d: gathered = the output gradient)
typedef Geo<0, 32, 64, 25, 25, 12, 12, 4, 4, 2, 1> T_mm_conv2;         // MultiMNIST features.2
typedef Geo<0, 32, 64, 25, 25, 12, 12, 5, 5, 2, 1> T_mm_convT3;        // hallucinate.6
typedef Geo<0, 64, 128, 12, 12, 6, 6, 4, 4, 2, 1> T_mm_conv3;          // features.5 and hallucinate.3

int try_launch_wgrad_tap(const WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx) {
    const int sel = mmvae_knob("wgrad_tap", 0);        // opt-in (1: every compiled layer, 2: hallucinate.6 only): DESIGN.md, tried and lost
    if (!sel) return 0;
    if (p.trA || p.trP || p.c.a_bcast_n > 0 || p.c.a_mask || p.c.a_affine || p.c.a_act != ACT_NONE || p.p_affine || p.p_act != ACT_NONE) return 0;
    int rc;
    //                       TG IB WAVES
    if ((rc = try_wt<T_mm_convT3, 13, 1, 4>(p, stream, ctx)) != 0) return rc;      // 2 tap groups (13 + 12)
    if (sel == 2) return 0;
    if ((rc = try_wt<T_mm_conv2, 8, 1, 4>(p, stream, ctx)) != 0) return rc;        // 2 tap groups x 2 channel tiles of N
    if ((rc = try_wt<T_mm_conv3, 4, 1, 4>(p, stream, ctx)) != 0) return rc;        // 4 tap groups x 4 channel tiles of N
    return 0;
}

int try_launch_convres_wgrad(WgradParams& p, hipStream_t stream, WgradSlabCtx* ctx) {
    {
        const int rc = try_launch_wgrad_tap(p, stream, ctx);
        if (rc != 0) return rc;
    }
    const bool forced = (p.trA && p.trA->kind) || (p.trP && p.trP->kind);      // staging transforms exist only here
    if (!mmvae_knob("convres_wgrad", 0) && !forced) return 0;       // opt-in: slower than the streamed kernel inside the step (see header)
    if (p.c.a_bcast_n > 0 || p.c.a_mask || p.c.a_affine || p.c.a_act != ACT_NONE || p.p_affine || p.p_act != ACT_NONE) {
        MMVAE_REQUIRE(!forced, "convres wgrad: unsupported operand options with a staging transform");
        return 0;
    }
    int rc;
    //                          NI WAVES NSPLIT
    if ((rc = try_crw<W_mm_convT3, 1, 8, 1>(p, stream, ctx)) != 0) return rc;
    if (mmvae_knob("convres_wgrad", 0) == 2 && !forced) return 0;          // (2: hallucinate.6 only, the one layer it wins alone)
    if ((rc = try_crw<W_mm_conv2, 1, 8, 1>(p, stream, ctx)) != 0) return rc;
    if ((rc = try_crw<W_mm_conv3, 1, 8, 2>(p, stream, ctx)) != 0) return rc;
    MMVAE_REQUIRE(!forced, "convres wgrad: no kernel is compiled for the geometry of a launch with a staging transform");
    return 0;
}
