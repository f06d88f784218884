// Compile-time geometry of the image-resident conv kernels (convres.hip).  Pure C++17 (no HIP): the same header is
// compiled by tests/test_cpu_convres_geo.py into a host program that checks every (class, row, tap) address against the
// gather semantics of gemm.h (GatherCommon / GatherClass).
//
// A layer is one gather GEMM over pixels in one of two forms (layers.h: plan_fwdform / plan_classform):
//   FORM 0 (fwd form)  : rows = pixels of the SMALL side [OH][OW]; tap (ty,tx) reads the BIG side at
//                        (oy*S - PAD + ty, ox*S - PAD + tx)                 (Conv2d forward, ConvTranspose2d data gradient)
//   FORM 1 (class form): rows = pixels of the BIG side [OH][OW] split into S*S stride-parity classes; tap (ty,tx) of class
//                        (ph,pw) reads the SMALL side at (jy + offy - ty, jx + offx - tx)
//                                                                            (ConvTranspose2d forward, Conv2d data gradient)
// The gathered image [AH][AW][C] of an image sits in LDS ONCE, zero ring included, laid out so that for a fixed tap the
// pixels of consecutive rows are consecutive LDS pixels (FORM 0: the x axis is de-interleaved into S parity planes) --
// every tap of every k-step is then `lane base + compile-time immediate`, and a pixel pitch of C*2+16 bytes makes the
// 16-byte fragment reads of 16 consecutive pixels hit 16 different 16-byte bank slots.
#pragma once

namespace crgeo {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }      // floor division, b > 0
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }

// PIXPAD_: bytes added to a pixel's C*2 (16: conflict-free 16-byte row reads of the forward / data-gradient kernels; 0 for the
// weight-gradient kernel, whose transposed 8-byte reads do not need it and whose two resident tensors need the room)
template <int FORM_, int C_, int N_, int AH_, int AW_, int OH_, int OW_, int KH_, int KW_, int S_, int PAD_, int PIXPAD_ = 16>
struct Geo {
    static constexpr int FORM = FORM_, C = C_, N = N_, AH = AH_, AW = AW_, OH = OH_, OW = OW_, KH = KH_, KW = KW_, S = S_, PAD = PAD_;
    static constexpr int NCLS = FORM == 0 ? 1 : S * S;
    static constexpr int PIX = C * 2 + PIXPAD_;            // LDS bytes per pixel
    static constexpr int KSTEP_PER_TAP = C / 16;           // 32x32x16 MFMA k-steps per tap
    static_assert(C % 8 == 0 && N % 32 == 0, "channel counts");      // (C % 16 == 0 for the MFMA k-steps; 8-channel slices: wgrad)

    // ---- per-class row grid / taps (identical to layers.h plan_fwdform / plan_classform)
    static constexpr int ph(int c) { return FORM == 0 ? 0 : c / S; }
    static constexpr int pw(int c) { return FORM == 0 ? 0 : c % S; }
    static constexpr int kh0(int c) { return (ph(c) + PAD) % S; }
    static constexpr int kw0(int c) { return (pw(c) + PAD) % S; }
    static constexpr int OY(int c) { return FORM == 0 ? OH : (OH - ph(c) + S - 1) / S; }
    static constexpr int OX(int c) { return FORM == 0 ? OW : (OW - pw(c) + S - 1) / S; }
    static constexpr int TH(int c) { return FORM == 0 ? KH : (KH - kh0(c) + S - 1) / S; }
    static constexpr int TW(int c) { return FORM == 0 ? KW : (KW - kw0(c) + S - 1) / S; }
    static constexpr int offy(int c) { return FORM == 0 ? -PAD : (ph(c) + PAD - kh0(c)) / S; }
    static constexpr int offx(int c) { return FORM == 0 ? -PAD : (pw(c) + PAD - kw0(c)) / S; }
    static constexpr int K(int c) { return TH(c) * TW(c) * C; }
    static constexpr int KSTEPS(int c) { return K(c) / 16; }
    static constexpr int OYX(int c) { return OY(c) * OX(c); }
    static constexpr int TPI(int c) { return cdiv(OYX(c), 32); }         // 32-row tiles per image
    // output pixel of class row (jy, jx)
    static constexpr int osy() { return FORM == 0 ? 1 : S; }
    static constexpr int out_y(int c, int jy) { return jy * osy() + ph(c); }
    static constexpr int out_x(int c, int jx) { return jx * osy() + pw(c); }
    // gathered pixel of (row, tap) in image coordinates (may be outside the image: zero)
    static constexpr int gy(int c, int jy, int ty) { return FORM == 0 ? jy * S - PAD + ty : jy + offy(c) - ty; }
    static constexpr int gx(int c, int jx, int tx) { return FORM == 0 ? jx * S - PAD + tx : jx + offx(c) - tx; }

    // ---- FORM 0 LDS image: rows [0, LR) <-> iy = row - PAD; x de-interleaved: ix = S*(col + DMIN) + plane
    static constexpr int DMIN = fdiv(-PAD, S), DMAX = fdiv(KW - 1 - PAD, S);
    static constexpr int PW0 = OW + DMAX - DMIN;                        // columns per plane
    static constexpr int LR0 = (OH - 1) * S + KH;                       // rows
    // ---- FORM 1 LDS image: ring of RT/RB rows and RL/RR columns around [AH][AW]
    static constexpr int ring_lo(bool x) {
        int r = 0;
        for (int c = 0; c < NCLS; ++c) r = cmax(r, x ? TW(c) - 1 - offx(c) : TH(c) - 1 - offy(c));
        return r;
    }
    static constexpr int ring_hi(bool x) {
        int r = 0;
        for (int c = 0; c < NCLS; ++c) r = cmax(r, x ? OX(c) - 1 + offx(c) - (AW - 1) : OY(c) - 1 + offy(c) - (AH - 1));
        return r;
    }
    static constexpr int RT = FORM == 1 ? ring_lo(false) : 0, RB = FORM == 1 ? ring_hi(false) : 0;
    static constexpr int RL = FORM == 1 ? ring_lo(true) : 0, RR = FORM == 1 ? ring_hi(true) : 0;
    static constexpr int AHP = RT + AH + RB, AWP = RL + AW + RR;
    static constexpr int IMG_BYTES = FORM == 0 ? LR0 * S * PW0 * PIX : AHP * AWP * PIX;

    // LDS byte offset of image pixel (iy, ix) (0 <= iy < AH, 0 <= ix < AW), or -1 when no row / tap ever reads it
    static constexpr int cell(int iy, int ix) {
        if (FORM == 0) {
            const int row = iy + PAD, plane = ix % S, col = ix / S - DMIN;
            if (row < 0 || row >= LR0 || col < 0 || col >= PW0) return -1;
            return ((row * S + plane) * PW0 + col) * PIX;
        }
        return ((iy + RT) * AWP + ix + RL) * PIX;
    }
    // lane base of class row (jy, jx) and the immediate of tap (ty, tx): base + tap_off = LDS offset of the gathered pixel
    static constexpr int row_stride(int) { return FORM == 0 ? S * S * PW0 * PIX : AWP * PIX; }   // per jy
    static constexpr int col_stride(int) { return PIX; }                                        // per jx
    static constexpr int base0(int c) {
        return FORM == 0 ? 0 : ((offy(c) + RT - (TH(c) - 1)) * AWP + offx(c) + RL - (TW(c) - 1)) * PIX;
    }
    static constexpr int tap_off(int c, int ty, int tx) {
        if (FORM == 0) {
            const int e = tx - PAD, d = fdiv(e, S), p = e - d * S;
            return ((ty * S + p) * PW0 + d - DMIN) * PIX;
        }
        return ((TH(c) - 1 - ty) * AWP + (TW(c) - 1 - tx)) * PIX;
    }
    // k-step kk (16 channels) of class c: tap and first channel
    static constexpr int step_tap(int kk) { return kk / KSTEP_PER_TAP; }
    static constexpr int step_c0(int kk) { return (kk % KSTEP_PER_TAP) * 16; }
    static constexpr int step_off(int c, int kk) {
        const int tap = step_tap(kk);
        return tap_off(c, tap / TW(c), tap % TW(c)) + step_c0(kk) * 2;
    }
    static constexpr int max_ksteps() { int r = 0; for (int c = 0; c < NCLS; ++c) r = cmax(r, KSTEPS(c)); return r; }
    static constexpr int max_tpi() { int r = 0; for (int c = 0; c < NCLS; ++c) r = cmax(r, TPI(c)); return r; }
    static constexpr int tab_entries() { int r = 0; for (int c = 0; c < NCLS; ++c) r += TPI(c) * 32; return r; }
    static constexpr int tab_base(int c) { int r = 0; for (int i = 0; i < c; ++i) r += TPI(i) * 32; return r; }
};

}  // namespace crgeo
