// Image-resident implicit-GEMM convolution kernels for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// gemm_gather_kernel (gemm.hip) streams one [128 rows][64 k] gathered tile per k-tile through LDS: every input pixel is
// fetched from L2 once per TAP that uses it (4x for k4/s2, 6.25x for k5/s2) and every fetch carries its own address
// arithmetic -- with N <= 64 output channels a staged row feeds 2-4 MFMAs and the kernel is bound by instruction issue and
// by what one CU can pull out of L2 (~70 GB/s), at 5-10 % of the MFMA rate.  Here a workgroup keeps the gathered images of
// NI samples RESIDENT in LDS (zero ring included, laid out by convres_geo.h so that a tap is `lane base + immediate`):
//   * each input byte crosses L2 -> CU once; the k-loop has no address arithmetic at all: one ds_read_b128 per operand
//     fragment, one MFMA per (row tile, column tile, k-step), fully unrolled with compile-time tap offsets;
//   * the packed weights [N][Kpad] of a stride-parity class come through LDS in chunks of CH k-steps shared by all waves;
//   * a wave owns up to MT row tiles (32 rows of the class, enumerated across the NI images) x NT column tiles (32
//     channels); accumulators persist across the weight chunks;
//   * the product is oriented [pixel][channel]: a lane owns ONE output channel, so every per-channel quantity of the
//     epilogue (bias, BatchNorm affine of the saved activation, mean / rstd, column sums) is one register, and the column
//     statistics are in-lane sums -- no cross-lane reduction, no trip through LDS;
//   * output rows are addressed through a per-workgroup table (row -> byte offset, out-of-range rows -> an offset beyond
//     the buffer that the hardware range check drops).
// Same GemmParams / epilogue semantics as gemm_gather_kernel for the features the conv layers use (raw bf16 store, column
// statistics, d-activation with BatchNorm-backward sums); launch_gemm_gather tries this path first and falls back.
#include "gemm.h"
#include "convres_geo.h"
#include "convres.h"
#include "bn_dev.h"
#include "convres_epi.h"
#include <type_traits>
#include <cstdlib>

namespace {


template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct ConvResArgs {
    const bf16* A;              // gathered tensor [nimg][AH][AW][C]
    int nimg, group_n;          // images in all / per BatchNorm group
    const bf16* Wp[4];          // packed weights per class [N][Kpad] (BDIR kernels: the fragment-major copy, PackDesc::frag)
    int Kpad[4];
    int Npad;
    bf16* out;                  // [nimg][OH][OW][ldo]
    int ldo;
    float2* colstats;           // [groups][SLOTS][N] += (sum v, sum v^2) or null
    const bf16* d_r;            // saved raw tensor of the output geometry (ld = ldo) or null
    const float2* d_affine;     // [groups][N] or null
    const float2* d_meanrstd;   // [groups][N] or null
    float2* d_red;              // [groups][SLOTS][N] += (sum x, sum x*xhat) or null
    int dbg;                    // measurement aid (mmvae_debug_set "convres_dbg"): 1 no output stores, 2 no MFMA loop, 4 no staging
    // ---- transform applied to the gathered tensor while it is staged (TR template parameter)
    // TR 1: A is a RAW conv output; staged value = Swish(BatchNorm(A)) with the tables made from `fin.stats` (training) or the
    //       running buffers; workgroup 0 also writes fin.affine / fin.meanrstd and updates the running statistics -- the
    //       job of bn_act_kernel, which then leaves the step's main chain
    BnFinalizeArgs fin;
    // TR 2: A is db (gradient w.r.t. a BatchNorm output); staged value = its BatchNorm-backward dr = gamma*rstd*(db - mean(db)
    //       - xhat*mean(db*xhat)); workgroup 0 adds the parameter gradients -- the job of bn_bwd_apply_kernel
    const bf16* t_r;            // raw tensor the BatchNorm normalised (geometry of A)
    const float2* t_red;        // [groups][SLOTS][C] (sum db, sum db*xhat)
    const float2* t_mr;         // [groups][C] (mean, rstd)
    const float* t_gamma;
    float* t_dgamma; float* t_dbeta;    // += (may be null)
    float t_inv_cnt;            // 1 / elements per channel per group
    int t_groups;
    bf16* stage_out;            // the staged tensor of a TR 1 / TR 2 launch is also written here (null: not)
    unsigned long long* ts;     // measurement aid: per-wave s_memrealtime stamps (100 MHz) [wg][wave][16] or null
};

constexpr unsigned OOB_OFF = 0x40000000u;

// row table of a workgroup: class c starts at tab_base (classes are padded to whole 32-row tiles)
template <class G, int NI>
constexpr int tab_base(int c) {
    int r = 0;
    for (int i = 0; i < c; ++i) r += crgeo::cdiv(NI * G::OYX(i), 32) * 32;
    return r;
}

// CH > 0: the packed weights of a class come through LDS in chunks of CH k-steps shared by all waves.
// CH == 0 ("direct B"): every wave streams the weight fragments of its own column tiles straight from L2 into registers, PD
// k-steps ahead (fragment-major packed weights: a fragment is two contiguous 512-byte runs).  No weight buffer, no barrier
// after the staging one: for the layers whose weights dwarf their activations (N >= 64, K >= 512 at 6x6 / 12x12 pixels) a
// chunk of weights was 0.25 us of MFMA work behind two barriers and an exposed L2 round trip.
// NSPLIT: the output channels are split over gridDim.y workgroups (direct-B only: each streams its own weight columns).
// KSPL: the k range of a tile is split over KSPL waves by tap ROW (partial tiles meet in LDS); for the layers with a handful of
// pixels per image and K = 1024..2048, where a workgroup's images give one or two row tiles.
template <class G, int NI, int CH, int WAVES, int NG, bool SCR_OWN, int NSPLIT = 1, int KSPL = 1>
struct CrLayout {
    static constexpr int NTHR = WAVES * 64;
    static constexpr int MG = WAVES / (NG * KSPL), NT = G::N / 32 / NSPLIT / NG;
    static constexpr int max_mt() { int m = 0; for (int c = 0; c < G::NCLS; ++c) m = crgeo::cmax(m, crgeo::cdiv(crgeo::cdiv(NI * G::OYX(c), 32), MG)); return m; }
    static constexpr int RED_BYTES = KSPL > 1 ? WAVES * max_mt() * NT * 4096 : 0;
    static constexpr bool BDIR = CH == 0;
    static constexpr int BP = (CH * 16 + 8) * 2;            // weight row pitch in LDS (bytes): odd multiple of 16
    static constexpr int W_BYTES = BDIR ? 0 : G::N * BP;
    static_assert(!BDIR || SCR_OWN, "direct-B kernels have no weight buffer to lend to the epilogue");
    static constexpr int IMG_ALL = NI * G::IMG_BYTES;
    static constexpr int TAB_BYTES = tab_base<G, NI>(G::NCLS) * 4;
    static constexpr int SCR_PITCH = 80, SCR_BYTES = 32 * SCR_PITCH;   // per-wave epilogue scratch [32 rows][64 B + 16]
    static constexpr int TRT_BYTES = G::C * 16;             // per-channel staging-transform coefficients
    static constexpr int OFF_W = IMG_ALL, OFF_TAB = OFF_W + W_BYTES, OFF_TRT = OFF_TAB + TAB_BYTES,
                         OFF_SCR = OFF_TRT + TRT_BYTES, OFF_RED = OFF_SCR + (SCR_OWN ? WAVES * SCR_BYTES : 0), TOTAL = OFF_RED + RED_BYTES;
    static_assert((NSPLIT == 1 && KSPL == 1) || BDIR, "column / k splits exist for the direct-B kernels only");
    static_assert(G::N % (32 * NSPLIT * NG) == 0 && WAVES % (NG * KSPL) == 0, "wave decomposition");
    static_assert(SCR_OWN || W_BYTES >= WAVES * SCR_BYTES, "the weight buffer doubles as the epilogue scratch");
    static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

// MODE 0: forward (raw bf16 store + column statistics); MODE 1: data gradient (d-Swish of the saved tensor + BatchNorm-backward sums)
template <class G, int NI, int CH, int WAVES, int NG, bool SCR_OWN, int MODE, int TR, int NSPLIT = 1, int KSPL = 1>
__global__ __launch_bounds__(WAVES * 64) void convres_kernel(const ConvResArgs a) {
    using L = CrLayout<G, NI, CH, WAVES, NG, SCR_OWN, NSPLIT, KSPL>;
    constexpr int NT = L::NT, MG = L::MG, NTHR = WAVES * 64;
    constexpr int BP = L::BP, SCR_PITCH = L::SCR_PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const img_s = smem;
    char* const w_s = smem + L::OFF_W;
    unsigned* const tab = reinterpret_cast<unsigned*>(smem + L::OFF_TAB);
    float4* const trt = reinterpret_cast<float4*>(smem + L::OFF_TRT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int kg = wave % KSPL, ng = (wave / KSPL) % NG, mg = wave / (KSPL * NG);
    const int ntb = blockIdx.y * (G::N / 32 / NSPLIT) + ng * NT;        // first 32-channel tile of this wave
    const int img0 = blockIdx.x * NI;
    const int grp = img0 / a.group_n;
    auto stamp = [&](int idx) {
        if (a.ts && lane == 0) a.ts[((size_t)blockIdx.x * WAVES + wave) * 16 + idx] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    // ---- weight chunks: global -> registers (issued a phase ahead of their use) -> LDS
    constexpr bool BDIR = L::BDIR;
    constexpr int WV = BDIR ? 1 : crgeo::cdiv(G::N * CH * 2, NTHR);        // 16-byte vectors per thread of a full chunk
    i32x4c wreg[WV];
    auto w_fetch = [&](auto ci, auto chi) {
        constexpr int c = decltype(ci)::value, ch = decltype(chi)::value;
        constexpr int CHK = crgeo::cmin(CH, G::KSTEPS(c) - ch * CH), VPR = CHK * 2 + (BDIR ? 1 : 0), NV = G::N * VPR;
        const bf16* wsrc = a.Wp[c] + ch * CH * 16;
        const int kpad = a.Kpad[c];
#pragma unroll
        for (int it = 0; it < crgeo::cdiv(NV, NTHR); ++it) {
            const int v = tid + it * NTHR;
            const int n = v / VPR, kv = v - n * VPR;
            if (v < NV) wreg[it] = *reinterpret_cast<const i32x4c*>(wsrc + (size_t)n * kpad + kv * 8);
        }
    };
    auto w_store = [&](auto ci, auto chi) {
        constexpr int c = decltype(ci)::value, ch = decltype(chi)::value;
        constexpr int CHK = crgeo::cmin(CH, G::KSTEPS(c) - ch * CH), VPR = CHK * 2 + (BDIR ? 1 : 0), NV = G::N * VPR;
#pragma unroll
        for (int it = 0; it < crgeo::cdiv(NV, NTHR); ++it) {
            const int v = tid + it * NTHR;
            const int n = v / VPR, kv = v - n * VPR;
            if (v < NV) *reinterpret_cast<i32x4c*>(w_s + n * BP + kv * 16) = wreg[it];
        }
    };
    if constexpr (!BDIR) w_fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

    // ---- image staging: batches of SB vectors per thread (loads of a batch in flight together)
    constexpr int VPP = G::C / 8;                               // 16-byte vectors per pixel
    constexpr int NVI = NI * G::AH * G::AW * VPP;
    constexpr int IT = crgeo::cdiv(NVI, NTHR), SB = 4, NB = crgeo::cdiv(IT, SB);
    static_assert(NTHR % VPP == 0, "a thread keeps its channel vector across staging iterations");
    const bf16* const src = a.A + (size_t)img0 * (G::AH * G::AW * G::C);
    const bf16* const src_r = TR == 2 ? a.t_r + (size_t)img0 * (G::AH * G::AW * G::C) : nullptr;
    bf16* const stage_dst = (TR != 0 && a.stage_out && blockIdx.y == 0) ? a.stage_out + (size_t)img0 * (G::AH * G::AW * G::C) : nullptr;
    i32x4c ireg[SB], rreg[TR == 2 ? SB : 1];
    auto i_fetch = [&](int b) {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int v = tid + (b * SB + i) * NTHR;
            if (b * SB + i < IT && v < NVI) {
                ireg[i] = *reinterpret_cast<const i32x4c*>(src + (size_t)v * 8);
                if constexpr (TR == 2) rreg[i] = *reinterpret_cast<const i32x4c*>(src_r + (size_t)v * 8);
            }
        }
    };
    float tc0[TR ? 8 : 1], tc1[TR ? 8 : 1], tc2[TR == 2 ? 8 : 1];     // per-channel coefficients of this thread's 8 channels
    auto i_store = [&](int b) {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int v = tid + (b * SB + i) * NTHR;
            if (b * SB + i < IT && v < NVI) {
                const int pix = v / VPP, cv = v - pix * VPP;
                const int img = pix / (G::AH * G::AW), p2 = pix - img * (G::AH * G::AW);
                const int iy = p2 / G::AW, ix = p2 - iy * G::AW;
                const int cell = G::cell(iy, ix);
                i32x4c val = ireg[i];
                if constexpr (TR == 1) {
                    const bf16x8 x = __builtin_bit_cast(bf16x8, val);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)swish_fast((float)x[j] * tc0[j] + tc1[j]);
                    val = __builtin_bit_cast(i32x4c, o);
                } else if constexpr (TR == 2) {
                    const bf16x8 x = __builtin_bit_cast(bf16x8, val), rr = __builtin_bit_cast(bf16x8, rreg[i]);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)x[j] * tc0[j] + ((float)rr[j] * tc1[j] + tc2[j]));
                    val = __builtin_bit_cast(i32x4c, o);
                }
                if (cell >= 0) *reinterpret_cast<i32x4c*>(img_s + img * G::IMG_BYTES + cell + cv * 16) = val;
                if constexpr (TR != 0)
                    if (stage_dst) *reinterpret_cast<i32x4c*>(stage_dst + (size_t)v * 8) = val;
            }
        }
    };
    if (!(a.dbg & 4)) i_fetch(0);

    // ---- zero the image area (rings), build the row table and the staging-transform coefficients of this group
    {
        const i32x4c z = {0, 0, 0, 0};
        for (int i = tid * 16; i < L::IMG_ALL; i += NTHR * 16) *reinterpret_cast<i32x4c*>(img_s + i) = z;
        const unsigned img_out = (unsigned)(G::OH * G::OW) * (unsigned)a.ldo * 2u;
        static_for<0, G::NCLS>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            constexpr int ROWS = NI * G::OYX(c), T = crgeo::cdiv(ROWS, 32);
            for (int e = tid; e < T * 32; e += NTHR) {
                unsigned v = OOB_OFF;
                if (e < ROWS) {
                    const int img = e / G::OYX(c), q = e - img * G::OYX(c);
                    const int jy = q / G::OX(c), jx = q - jy * G::OX(c);
                    v = (unsigned)img * img_out + (unsigned)((G::out_y(c, jy) * G::OW + G::out_x(c, jx)) * a.ldo * 2);
                }
                tab[tab_base<G, NI>(c) + e] = v;
            }
        });
        if constexpr (TR == 1) {
            for (int ch = tid; ch < G::C; ch += NTHR) {
                float2 aff, mr;
                bn_channel_tables(a.fin, grp, ch, aff, mr);
                trt[ch] = make_float4(aff.x, aff.y, 0.f, 0.f);
            }
            if (blockIdx.x == 0 && blockIdx.y == 0) {       // the tables backward reads, the running statistics: once per layer
                for (int i = tid; i < a.fin.G * G::C; i += NTHR) {
                    float2 aff, mr;
                    bn_channel_tables(a.fin, i / G::C, i % G::C, aff, mr);
                    a.fin.affine[i] = aff; a.fin.meanrstd[i] = mr;
                }
                bn_running_update(a.fin, tid, NTHR);
            }
        } else if constexpr (TR == 2) {
            for (int ch = tid; ch < G::C; ch += NTHR) {
                float sx = 0.f, sy = 0.f;
                for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
                    const float2 t = a.t_red[((size_t)grp * MMVAE_STAT_SLOTS + q) * G::C + ch];
                    sx += t.x; sy += t.y;
                }
                const float2 mr = a.t_mr[grp * G::C + ch];
                const float g = a.t_gamma[ch] * mr.y, m1 = sx * a.t_inv_cnt, m2 = sy * a.t_inv_cnt;
                // dr = g*(db - m1 - (r - mean)*rstd*m2) = g*db + (-g*m2*rstd)*r + (g*m2*rstd*mean - g*m1)
                const float cb = -g * m2 * mr.y;
                trt[ch] = make_float4(g, cb, -cb * mr.x - g * m1, 0.f);
            }
            if (blockIdx.x == 0 && blockIdx.y == 0 && (a.t_dgamma || a.t_dbeta)) {
                for (int ch = tid; ch < G::C; ch += NTHR) {
                    float tg = 0.f, tb = 0.f;
                    for (int gg = 0; gg < a.t_groups; ++gg)
                        for (int q = 0; q < MMVAE_STAT_SLOTS; ++q) {
                            const float2 t = a.t_red[((size_t)gg * MMVAE_STAT_SLOTS + q) * G::C + ch];
                            tb += t.x; tg += t.y;
                        }
                    if (a.t_dgamma) a.t_dgamma[ch] += tg;
                    if (a.t_dbeta) a.t_dbeta[ch] += tb;
                }
            }
        }
    }
    stamp(1);
    __syncthreads();
    stamp(2);
    if constexpr (TR != 0) {
        const int cv0 = (tid % VPP) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 t = trt[cv0 + j];
            tc0[j] = t.x; tc1[j] = t.y;
            if constexpr (TR == 2) tc2[j] = t.z;
        }
    }
    if (!(a.dbg & 4)) {
        i_store(0);
        for (int b = 1; b < NB; ++b) { i_fetch(b); i_store(b); }
    }
    if constexpr (!BDIR) w_store(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    stamp(3);
    __syncthreads();
    stamp(4);

    // ---- epilogue constants of this lane's channels
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out, 0, (int)((size_t)a.nimg * G::OH * G::OW * a.ldo * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.d_r ? a.d_r : a.out), 0, (int)((size_t)a.nimg * G::OH * G::OW * a.ldo * 2), 0x00020000);
    const unsigned wg_out = (unsigned)img0 * (unsigned)(G::OH * G::OW) * (unsigned)a.ldo * 2u;
    float dsc[NT], dsh[NT], dmean[NT], drstd[NT], s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (ntb + nt) * 32 + r;
        dsc[nt] = 1.f; dsh[nt] = 0.f; dmean[nt] = 0.f; drstd[nt] = 0.f; s1[nt] = 0.f; s2[nt] = 0.f;
        if (MODE == 1 && a.d_affine) { const float2 t = a.d_affine[grp * G::N + co]; dsc[nt] = t.x; dsh[nt] = t.y; }
        if (MODE == 1 && a.d_meanrstd) { const float2 t = a.d_meanrstd[grp * G::N + co]; dmean[nt] = t.x; drstd[nt] = t.y; }
    }
    char* const scr = (SCR_OWN ? smem + L::OFF_SCR : w_s) + wave * L::SCR_BYTES;
    __amdgpu_buffer_rsrc_t wrsrc[BDIR ? G::NCLS : 1];
    if constexpr (BDIR) {
#pragma unroll
        for (int i = 0; i < G::NCLS; ++i)
            wrsrc[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.Wp[i]), 0, a.Npad * a.Kpad[i] * 2, 0x00020000);
    }

    // ---- classes.  At the top of a chunk its weights are in LDS and a barrier has been passed.
    static_for<0, G::NCLS>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        constexpr int ROWS = NI * G::OYX(c), T = crgeo::cdiv(ROWS, 32), MT = crgeo::cdiv(T, MG);
        constexpr int KS = G::KSTEPS(c), CHE = BDIR ? KS : CH, NCH = crgeo::cdiv(KS, CHE);
        constexpr int KSL = KS / KSPL;                       // k-steps of this wave
        // this wave's row tiles: mg, mg + MG, ...
        int abase[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int t = mg + m * MG;
            int R = (t < T ? t : 0) * 32 + r;
            R = R < ROWS ? R : ROWS - 1;
            const int img = R / G::OYX(c), q = R - img * G::OYX(c);
            const int jy = q / G::OX(c), jx = q - jy * G::OX(c);
            abase[m] = img * G::IMG_BYTES + G::base0(c) + jy * G::row_stride(c) + jx * G::col_stride(c) + h * 16;
            if constexpr (KSPL > 1) {                        // this wave's tap rows: a constant offset (both forms are linear in ty)
                static_assert(G::TH(c) % KSPL == 0, "the k split is by tap row");
                constexpr int KSL_ = KS / KSPL;
                abase[m] += kg * (G::step_off(c, KSL_) - G::step_off(c, 0));
            }
        }
        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[m][nt][j] = 0.f;
        const int wrow = ((ng * NT) * 32 + r) * BP + h * 16;

        if constexpr (BDIR) {
            // weight fragments straight from L2, PD k-steps ahead; activation fragments from LDS one step ahead
            constexpr int PD = KSL < 8 ? KSL : 8;
            static_assert(KSPL == 1 || KSL % 2 == 0, "a wave's k range starts on a 32-column block of the packed weights");
            unsigned voff[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                voff[nt] = (unsigned)(((2 * (ntb + nt) + (r >> 4)) * (a.Kpad[c] >> 5) * 64 + h * 16 + (r & 15)) * 16) +
                           (unsigned)(kg * (KSL / 2) * 1024);
            auto bload = [&](auto ki, bf16x8 (&fb)[NT]) {
                constexpr int kk = decltype(ki)::value;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fb[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc[c], (int)voff[nt], (kk >> 1) * 1024 + (kk & 1) * 512, 0));
            };
            auto aload = [&](auto ki, bf16x8 (&fa)[MT]) {
                constexpr int kk = decltype(ki)::value;
                constexpr int aoff = G::step_off(c, kk);
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(img_s + abase[m] + aoff);
            };
            bf16x8 bq[PD][NT], af[MT];
            static_for<0, PD>([&](auto pi) { bload(pi, bq[decltype(pi)::value]); });
            aload(std::integral_constant<int, 0>{}, af);
            __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
            if (!(a.dbg & 2))
            static_for<0, KSL>([&](auto ki) {
                constexpr int kk = decltype(ki)::value;
                constexpr int KS = KSL;                      // (the loop below runs over this wave's steps)
                bf16x8 bcur[NT], an[MT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bcur[nt] = bq[kk % PD][nt];
                if constexpr (kk + PD < KS) bload(std::integral_constant<int, kk + PD>{}, bq[kk % PD]);
                if constexpr (kk + 1 < KS) aload(std::integral_constant<int, kk + 1>{}, an);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bcur[nt], acc[m][nt], 0, 0, 0);
                if constexpr (kk + PD < KS) __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
                if constexpr (kk + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
                if constexpr (kk + 1 < KS) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[m] = an[m];
                }
            });
            if constexpr (KSPL > 1) {
                // partial tiles of the KSPL waves of a tile meet in LDS; wave kg == 0 carries on with the sum
                float* const red = reinterpret_cast<float*>(smem + L::OFF_RED);
                constexpr int WSTR = L::max_mt() * NT * 1024;            // floats per wave
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int v4 = 0; v4 < 4; ++v4)
                            *reinterpret_cast<f32x4*>(red + wave * WSTR + ((m * NT + nt) * 64 + lane) * 16 + v4 * 4) =
                                f32x4{acc[m][nt][4 * v4], acc[m][nt][4 * v4 + 1], acc[m][nt][4 * v4 + 2], acc[m][nt][4 * v4 + 3]};
                __syncthreads();
                if (kg == 0) {
#pragma unroll
                    for (int k2 = 1; k2 < KSPL; ++k2)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int v4 = 0; v4 < 4; ++v4) {
                                    const f32x4 t = *reinterpret_cast<const f32x4*>(red + (wave + k2) * WSTR + ((m * NT + nt) * 64 + lane) * 16 + v4 * 4);
                                    acc[m][nt][4 * v4] += t[0]; acc[m][nt][4 * v4 + 1] += t[1];
                                    acc[m][nt][4 * v4 + 2] += t[2]; acc[m][nt][4 * v4 + 3] += t[3];
                                }
                }
                __syncthreads();
            }
        }
        static_for<0, NCH>([&](auto chi) {
            constexpr int ch = decltype(chi)::value;
            constexpr int CHK = crgeo::cmin(CHE, KS - ch * CHE);         // k-steps in this chunk
            constexpr bool last_chunk = ch + 1 == NCH;
            constexpr bool has_next = !BDIR && (!last_chunk || c + 1 < G::NCLS);
            constexpr int cn = last_chunk ? c + 1 : c, chn = last_chunk ? 0 : ch + 1;
            if constexpr (has_next) w_fetch(std::integral_constant<int, cn>{}, std::integral_constant<int, chn>{});
            if (!BDIR && !(a.dbg & 2)) {
                // fragments of step kk+1 are requested before the MFMAs of step kk
                bf16x8 af[MT], bfr[NT];
                auto frag = [&](auto ki, bf16x8 (&fa)[MT], bf16x8 (&fb)[NT]) {
                    constexpr int kk = decltype(ki)::value;
                    constexpr int aoff = G::step_off(c, ch * CHE + kk);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        fb[nt] = *reinterpret_cast<const bf16x8*>(w_s + wrow + nt * 32 * BP + kk * 32);
#pragma unroll
                    for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(img_s + abase[m] + aoff);
                };
                frag(std::integral_constant<int, 0>{}, af, bfr);
                __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                static_for<0, CHK>([&](auto ki) {
                    constexpr int kk = decltype(ki)::value;
                    bf16x8 an[MT], bn[NT];
                    if constexpr (kk + 1 < CHK) frag(std::integral_constant<int, kk + 1>{}, an, bn);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[nt], acc[m][nt], 0, 0, 0);
                    if constexpr (kk + 1 < CHK) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
                    if constexpr (kk + 1 < CHK) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bfr[nt] = bn[nt];
                    }
                });
            }
            if constexpr (last_chunk) {
                stamp(5 + 2 * c);
                // ---- epilogue (cr_epilogue_tile): accumulator layout lane = channel, register j = row (j&3) + 8*(j>>2) + 4*h
                if constexpr (!SCR_OWN) __syncthreads();            // the scratch is the weight buffer: everybody is done reading it
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int t = mg + m * MG;
                    if (t < T && kg == 0) {
                        const unsigned* trow = tab + tab_base<G, NI>(c) + t * 32;
                        const unsigned off0 = trow[lane & 15], off1 = trow[16 + (lane & 15)];
                        const int rows_left = ROWS - t * 32;                  // < 32: the class's last, partial tile
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const unsigned cob = (unsigned)(((ntb + nt) * 32 + (lane >> 4) * 8) * 2) + wg_out;
                            if (rows_left >= 32)
                                cr_epilogue_tile<MODE, false>(acc[m][nt], scr, lane, 32, off0 + cob, off1 + cob, orsrc, rrsrc,
                                                              dsc[nt], dsh[nt], dmean[nt], drstd[nt], s1[nt], s2[nt], !(a.dbg & 1));
                            else
                                cr_epilogue_tile<MODE, true>(acc[m][nt], scr, lane, rows_left, off0 + cob, off1 + cob, orsrc, rrsrc,
                                                             dsc[nt], dsh[nt], dmean[nt], drstd[nt], s1[nt], s2[nt], !(a.dbg & 1));
                        }
                    }
                }
            }
            if constexpr (last_chunk) stamp(6 + 2 * c);
            if constexpr (has_next) {
                __syncthreads();                                        // every wave is done with this chunk (and the scratch)
                w_store(std::integral_constant<int, cn>{}, std::integral_constant<int, chn>{});
                __syncthreads();
            }
        });
    });

    stamp(13);
    // ---- column sums: one atomic pair per channel per wave
    float2* red = MODE == 1 ? a.d_red : a.colstats;
    if (red && kg == 0) {
        const int slot = (int)(blockIdx.x + wave) % MMVAE_STAT_SLOTS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float t1 = s1[nt] + __shfl_xor(s1[nt], 32, 64), t2 = s2[nt] + __shfl_xor(s2[nt], 32, 64);
            if (h == 0) {
                float2* d = red + ((size_t)grp * MMVAE_STAT_SLOTS + slot) * G::N + (ntb + nt) * 32 + r;
                atomicAdd(&d->x, t1);
                atomicAdd(&d->y, t2);
            }
        }
    }
    stamp(14);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
namespace {

// does the runtime gather plan equal the compile-time geometry G?
template <class G>
bool geo_matches(const GemmParams& p) {
    const GatherCommon& c = p.c;
    if (c.C != G::C || c.Ald != G::C || c.N != G::N || c.AH != G::AH || c.AW != G::AW || c.OH != G::OH || c.OW != G::OW) return false;
    if (c.nclasses != G::NCLS) return false;
    if (G::FORM == 0) {
        if (c.sy != G::S || c.sx != G::S || c.dy != 1 || c.dx != 1 || c.osy != 1 || c.osx != 1) return false;
    } else {
        if (c.sy != 1 || c.sx != 1 || c.dy != -1 || c.dx != -1 || c.osy != G::S || c.osx != G::S) return false;
    }
    for (int i = 0; i < G::NCLS; ++i) {
        const GatherClass& k = p.cls[i];
        if (k.OY != G::OY(i) || k.OX != G::OX(i) || k.TH != G::TH(i) || k.TW != G::TW(i) || k.offy != G::offy(i) ||
            k.offx != G::offx(i) || k.ooy != G::ph(i) || k.oox != G::pw(i) || k.K != G::K(i) || k.Kpad < G::K(i))
            return false;
    }
    return true;
}

template <class G, int NI, int CH, int WAVES, int NG, bool SCR_OWN, int MODE, int TR, int NSPLIT, int KSPL>
int launch_cr_mode(const ConvResArgs& a, hipStream_t stream) {
    using L = CrLayout<G, NI, CH, WAVES, NG, SCR_OWN, NSPLIT, KSPL>;
    static std::atomic<unsigned> attr_set{0};
    if (mmvae_first_use_on_device(attr_set))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&convres_kernel<G, NI, CH, WAVES, NG, SCR_OWN, MODE, TR, NSPLIT, KSPL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    MMVAE_LAUNCH((convres_kernel<G, NI, CH, WAVES, NG, SCR_OWN, MODE, TR, NSPLIT, KSPL>), dim3(a.nimg / NI, NSPLIT), dim3(WAVES * 64), L::TOTAL, stream, a);
    MMVAE_TRY(mmvae_check_launch("convres"));
    return 1;
}

template <class G, int NI, int CH, int WAVES, int NG, bool SCR_OWN, int NSPLIT = 1, int KSPL = 1>
int launch_cr(const GemmParams& p, hipStream_t stream) {
    const GatherCommon& c = p.c;
    const int nimg = c.groups * c.group_n;
    ConvResArgs a{};
    a.A = c.A; a.nimg = nimg; a.group_n = c.group_n;
    constexpr bool BDIR = CH == 0;
    for (int i = 0; i < G::NCLS; ++i) {
        a.Wp[i] = BDIR ? p.cls[i].Wf : p.cls[i].Wp; a.Kpad[i] = p.cls[i].Kpad;
        if (BDIR && (!a.Wp[i] || a.Kpad[i] % 32 != 0)) return 0;      // no fragment-major copy of these weights: not this kernel
    }
    a.Npad = p.npad;
    a.out = p.out_bf; a.ldo = p.ldo; a.colstats = p.colstats;
    a.dbg = mmvae_knob("convres_dbg", 0);
    a.ts = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)mmvae_knob("convres_ts_hi", 0) << 32) |
                                                 (unsigned)mmvae_knob("convres_ts_lo", 0));
    a.d_r = p.d_r; a.d_affine = p.d_affine; a.d_meanrstd = p.d_meanrstd; a.d_red = p.d_red;
    const int kind = p.tr ? p.tr->kind : 0;
    MMVAE_REQUIRE(!(p.tr && p.tr->out == c.A) || NSPLIT == 1, "convres: an in-place staging by-product needs one workgroup per image tile");
    if (mmvae_probe_on()) {      // FLOPs the way the reference's FlopCounterMode counts the layer: cropped taps not subtracted
        char tag[96];
        snprintf(tag, sizeof(tag), "convres form%d %d>%d %dx%d>%dx%d k%d s%d %s tr%d img%d", G::FORM, G::C, G::N, G::AH, G::AW, G::OH, G::OW,
                 G::KH, G::S, p.d_r ? "dgrad" : "fwd", kind, nimg);
        mmvae_probe_tag(tag, 2.0 * (G::FORM == 0 ? G::OH * G::OW : G::AH * G::AW) * G::KH * G::KW * G::C * G::N * (double)nimg);
    }
    if (kind == 1) {
        a.fin = p.tr->fin; a.stage_out = p.tr->out;
        MMVAE_REQUIRE(a.fin.C == G::C && a.fin.G == c.groups && a.fin.affine && a.fin.meanrstd && a.fin.gamma && a.fin.beta,
                      "convres: BatchNorm tables of the staged operand do not match the layer");
        MMVAE_REQUIRE(!p.d_r, "convres: forward-type staging transform on a data-gradient launch");
        return launch_cr_mode<G, NI, CH, WAVES, NG, SCR_OWN, 0, 1, NSPLIT, KSPL>(a, stream);
    }
    if (kind == 2) {
        const GatherTransform& t = *p.tr;
        MMVAE_REQUIRE(p.d_r && t.r && t.red && t.mr && t.gamma && t.groups == c.groups, "convres: BatchNorm-backward staging needs r / sums / tables");
        a.t_r = t.r; a.t_red = t.red; a.t_mr = t.mr; a.t_gamma = t.gamma; a.t_dgamma = t.dgamma; a.t_dbeta = t.dbeta;
        a.t_inv_cnt = t.inv_cnt; a.t_groups = t.groups; a.stage_out = t.out;
        return launch_cr_mode<G, NI, CH, WAVES, NG, SCR_OWN, 1, 2, NSPLIT, KSPL>(a, stream);
    }
    return p.d_r ? launch_cr_mode<G, NI, CH, WAVES, NG, SCR_OWN, 1, 0, NSPLIT, KSPL>(a, stream)
                 : launch_cr_mode<G, NI, CH, WAVES, NG, SCR_OWN, 0, 0, NSPLIT, KSPL>(a, stream);
}

// the features of GemmParams this path implements
bool features_ok(const GemmParams& p) {
    const GatherCommon& c = p.c;
    if (!p.out_bf || p.out_f || p.out_act_bf || p.e_mask || p.d_mask || p.d_colsum || p.d_cmod > 0 || p.d_bcast_n > 0) return false;
    if (c.a_bcast_n > 0 || c.a_mask || c.a_affine || c.a_act != ACT_NONE) return false;
    if (p.ldo != c.N || p.bias) return false;
    if (p.d_r && (p.d_ld != p.ldo || p.d_act != ACT_SWISH)) return false;
    if (p.colstats && p.d_r) return false;
    if ((size_t)c.groups * c.group_n * c.OH * c.OW * p.ldo * 2 >= (size_t)OOB_OFF) return false;
    return true;
}

template <class G, int NI, int CH, int WAVES, int NG, bool SCR_OWN, int NSPLIT = 1, int KSPL = 1>
int try_cr(const GemmParams& p, hipStream_t stream) {
    if (!geo_matches<G>(p)) return 0;
    if (p.c.group_n % NI != 0) return 0;
    return launch_cr<G, NI, CH, WAVES, NG, SCR_OWN, NSPLIT, KSPL>(p, stream);
}

using crgeo::Geo;
//                FORM C    N   AH  AW  OH  OW  KH KW S  PAD
typedef Geo<0, 32, 64, 25, 25, 12, 12, 4, 4, 2, 1> G_mm_conv2;         // MultiMNIST features.2 forward
typedef Geo<0, 64, 128, 12, 12, 6, 6, 4, 4, 2, 1> G_mm_conv3;          // features.5 forward == hallucinate.3 data gradient
typedef Geo<1, 128, 64, 6, 6, 12, 12, 4, 4, 2, 1> G_mm_convT2;         // hallucinate.3 forward == features.5 data gradient
typedef Geo<1, 64, 32, 12, 12, 25, 25, 5, 5, 2, 1> G_mm_convT3;        // hallucinate.6 forward
typedef Geo<1, 64, 32, 12, 12, 25, 25, 4, 4, 2, 1> G_mm_conv2d;        // features.2 data gradient
typedef Geo<0, 32, 64, 25, 25, 12, 12, 5, 5, 2, 1> G_mm_convT3d;       // hallucinate.6 data gradient
typedef Geo<1, 256, 128, 2, 2, 6, 6, 4, 4, 2, 0> G_mm_convT1;          // hallucinate.0 forward == features.8 data gradient
typedef Geo<0, 128, 256, 6, 6, 2, 2, 4, 4, 2, 0> G_mm_conv4;           // features.8 forward == hallucinate.0 data gradient
// CelebA (64x64 images; celeba/model.py:101-150) -- the 16x16 / 8x8 pairs are also COCO's features.2 / hallucinate.6
typedef Geo<0, 32, 64, 32, 32, 16, 16, 4, 4, 2, 1> G_ca_conv2;         // features.2 forward == hallucinate.6 data gradient
typedef Geo<0, 64, 128, 16, 16, 8, 8, 4, 4, 2, 1> G_ca_conv3;          // features.5 forward == hallucinate.3 data gradient
typedef Geo<1, 128, 64, 8, 8, 16, 16, 4, 4, 2, 1> G_ca_convT2;         // hallucinate.3 forward == features.5 data gradient
typedef Geo<1, 64, 32, 16, 16, 32, 32, 4, 4, 2, 1> G_ca_convT3;        // hallucinate.6 forward == features.2 data gradient

}  // namespace

int try_launch_convres(const GemmParams& p, hipStream_t stream) {
    const bool forced = p.tr && p.tr->kind != 0;        // a staging transform exists only here: no fallback
    if ((!mmvae_knob("convres", 1) && !forced) || !features_ok(p)) {
        MMVAE_REQUIRE(!forced, "convres: a staging transform was requested for a launch this path does not cover");
        return 0;
    }
    int rc;
    const int nimg = p.c.groups * p.c.group_n;
    const int alt = mmvae_knob("convres_alt", 0);       // measurement aid: alternative tile configurations
    //                          NI  CH  WAVES NG SCR_OWN
    if ((rc = try_cr<G_mm_conv2, 1, 16, 8, 2, true>(p, stream)) != 0) return rc;
    if (alt != 3) {       // CH = 0: direct-B kernels (need the fragment-major weight copy; else the LDS-chunk form below)
        if ((rc = try_cr<G_mm_conv3, 2, 0, 8, 4, true>(p, stream)) != 0) return rc;
        if (nimg <= 256) { if ((rc = try_cr<G_mm_convT2, 2, 0, 8, 2, true>(p, stream)) != 0) return rc; }
        if ((rc = try_cr<G_mm_convT2, 4, 0, 8, 2, true>(p, stream)) != 0) return rc;
    }
    if ((rc = try_cr<G_mm_conv3, 2, 8, 8, 4, true>(p, stream)) != 0) return rc;
    if (nimg <= 256) { if ((rc = try_cr<G_mm_convT2, 2, 8, 4, 1, true>(p, stream)) != 0) return rc; }
    if ((rc = try_cr<G_mm_convT2, 4, 16, 8, 2, true>(p, stream)) != 0) return rc;
    if (alt == 1) { if ((rc = try_cr<G_mm_convT3, 4, 20, 8, 1, false>(p, stream)) != 0) return rc; }
    if ((rc = try_cr<G_mm_convT3, 2, 16, 4, 1, false>(p, stream)) != 0) return rc;
    if ((rc = try_cr<G_mm_conv2d, 1, 16, 8, 1, true>(p, stream)) != 0) return rc;
    if (alt != 4) {       // the 2x2 <-> 6x6 bottleneck layers (1 MB of weights): output channels over 4 workgroups, k over the waves
        //                          NI CH WAVES NG SCR_OWN NSPLIT KSPL
        if (nimg > 256 && alt != 5) { if ((rc = try_cr<G_mm_convT1, 16, 0, 5, 1, true, 4, 1>(p, stream)) != 0) return rc; }   // one round of 192 workgroups
        if ((rc = try_cr<G_mm_convT1, 8, 0, 6, 1, true, 4, 2>(p, stream)) != 0) return rc;
        if ((rc = try_cr<G_mm_conv4, 8, 0, 8, 2, true, 4, 4>(p, stream)) != 0) return rc;
    }
    if ((rc = try_cr<G_mm_convT3d, 2, 10, 8, 2, false>(p, stream)) != 0) return rc;
    if (mmvae_knob("convres_celeba", 1)) {
        if ((rc = try_cr<G_ca_conv2, 1, 16, 8, 2, true>(p, stream)) != 0) return rc;
        if ((rc = try_cr<G_ca_conv3, 2, 0, 8, 4, true>(p, stream)) != 0) return rc;
        if ((rc = try_cr<G_ca_convT2, 4, 0, 8, 2, true>(p, stream)) != 0) return rc;
        if ((rc = try_cr<G_ca_convT3, 2, 16, 4, 1, false>(p, stream)) != 0) return rc;
        // (the stride-1 8x8 <-> 5x5 pair, 1 MB of weights against 17 KB images, was tried here with 4 images per workgroup, a quarter
        //  of the channels and the k range split over wave pairs: 213 us for its two launches against 190 on gemm_gather -- each
        //  workgroup streams 512 KB of weights for 100 pixel rows; that layer wants the weights, not the images, resident.
        //  With the weights through LDS chunks shared by 8 waves -- G_ca_conv4 <4, 8, 8, 8, false>, G_ca_convT1 <2, 5, 8, 4, false> --
        //  the four launches of the pair take 440 us in the step against 447 on gemm_gather: 210-330 TFLOP/s either way)
    }
    MMVAE_REQUIRE(!forced, "convres: no kernel is compiled for the geometry of a launch with a staging transform");
    return 0;
}
