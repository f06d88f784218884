// Compile-time geometry of the ring-staged weight-gradient kernel (wgrad_ring.hip).  Pure C++17 (no HIP): the same header is
// compiled into a host program (tests/host/wgrad_geo_check.cpp) that replays the DMA fill of an LDS slot and every fragment
// address of every (class, pixel row, tap) against the definition of the convolution's weight gradient.
//
//   dW[n][(ty, tx)][c] = sum over images and pixels (oy, ox) of  S[img][oy][ox][n] * B[img][oy*ST - PAD + ty][ox*ST - PAD + tx][c]
//
// S is the tensor on the SMALL side of the layer ([OH][OW][N]: Conv2d -> the gradient of the raw output, ConvTranspose2d -> the
// activated input), B the tensor on the BIG side ([AH][AW][C]).  The taps split into ST*ST stride-parity classes: class
// (cy, cx) holds the taps with (ty - PAD) mod ST == cy, (tx - PAD) mod ST == cx, and all of them read ONE parity plane of B
// (rows iy == cy, columns ix == cx mod ST) at unit stride:  plane pixel (oy + dy, ox + dx), dy = (ty - PAD - cy) / ST.
// A workgroup owns one class (x a slice of NS of the N channels): per image it needs the small image and a quarter of the big
// one, and for a fixed tap consecutive pixel rows are consecutive plane cells, so a tap is a compile-time cell offset.
//
// LDS slot (one batch of IB images), everything in 64-byte granules = 32 channels of one pixel:
//   small region  [NTN = NS/32][RPAD rows][64 B]            rows = (image, oy, ox) of the batch, padded to whole k-steps of 16
//   big region    [CT = C/32][NCELLP cells][64 B]           cells = (image, r, c) of the class plane incl. its zero ring
// Four consecutive rows / cells of one channel tile are 256 contiguous bytes: the transposed 8-byte fragment reads
// (ds_read_b64_tr_b16, 4 k-rows x 64 B per 32-lane half) are bank-conflict free without padding or swizzle -- which the
// linear destination of an LDS-DMA (wave base + lane*16) could not express anyway.
#pragma once

namespace wrgeo {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int rup(int a, int b) { return cdiv(a, b) * b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

struct Src {            // where a 16-byte chunk of the slot comes from
    int tensor;         // 0: small-side tensor, 1: big-side tensor, -1: nothing (zero fill)
    int off;            // byte offset from (tensor base + first image of the batch [+ the channel slice for the small side])
};

// WGQ_: workgroups of a launch in quarters of the CU count (0: the launcher's default) -- layers with a large dW want few image
// groups, every group is one more partial copy of dW
// TGN_: the tap rows of a parity class are split over TGN workgroup classes (all of them read the class's plane): a stride-1 layer
// has ONE parity class of KH*KW taps, more accumulators than a workgroup's registers hold
// PAIR_: a workgroup of 2*WAVES waves takes TWO classes through ONE fill of the small image (waves 0..WAVES-1 the heavier class of
// the pair, the others the lighter one): the small image crosses the fabric twice per sample instead of four times, and every
// SIMD holds two waves, one of each class
template <int C_, int N_, int AH_, int AW_, int OH_, int OW_, int KH_, int KW_, int ST_, int PAD_, int NS_, int IB_, int WAVES_, int SLOTS_, int WGQ_ = 0,
          int TGN_ = 1, int PAIR_ = 0>
struct Geo {
    static constexpr int WGQ = WGQ_, TGN = TGN_, PAIR = PAIR_;
    static constexpr int C = C_, N = N_, AH = AH_, AW = AW_, OH = OH_, OW = OW_, KH = KH_, KW = KW_, ST = ST_, PAD = PAD_;
    static constexpr int NS = NS_, IB = IB_, WAVES = WAVES_, SLOTS = SLOTS_;
    static_assert(C % 32 == 0 && N % NS == 0 && NS % 32 == 0, "channel tiles are 32 wide");
    static constexpr int NCLS = ST * ST * TGN;                // workgroup classes: (parity class, tap-row group)
    static constexpr int NTN = NS / 32, CT = C / 32, MS = N / NS;
    static_assert(NTN == 1 || NTN == 2 || NTN == 4, "a wave owns one row tile of the slice: NS/32 must divide the wave count");
    static_assert(WAVES % NTN == 0, "wave grid");
    static constexpr int WC = WAVES / NTN;                    // waves side by side over the column tiles
    static constexpr int OYX = OH * OW, ROWS = IB * OYX, RPAD = rup(ROWS, 16), KST = RPAD / 16;

    // ---- one axis: tap t -> parity and plane shift
    static constexpr int par(int t) { return ((t - PAD) % ST + ST) % ST; }
    static constexpr int shift(int t) { return (t - PAD - par(t)) / ST; }          // exact: (t - PAD - par) is a multiple of ST
    static constexpr int t0(int p) { return (p + PAD) % ST; }                        // first tap of parity p; the others follow at t0 + k*ST
    static constexpr int ntap(int p, int K) { return t0(p) < K ? (K - 1 - t0(p)) / ST + 1 : 0; }
    static constexpr int dmin(int p) { return shift(t0(p)); }                        // shifts of a parity are consecutive: dmin + k
    // ---- class c = (cy*ST + cx)*TGN + tap-row group
    static constexpr int cy(int c) { return c / TGN / ST; }
    static constexpr int cx(int c) { return c / TGN % ST; }
    static constexpr int tg(int c) { return c % TGN; }
    static constexpr int NTY(int c) { return ntap(cy(c), KH) / TGN; }                // (tests/host/wgrad_geo_check.cpp: the split is even)
    static constexpr int KY0(int c) { return tg(c) * NTY(c); }                       // first tap row of the group inside its parity class
    static constexpr int NTX(int c) { return ntap(cx(c), KW); }
    static constexpr int NTAPS(int c) { return NTY(c) * NTX(c); }
    static constexpr int NCT(int c) { return NTAPS(c) * CT; }                        // column tiles (tap, channel tile)
    static constexpr int CPW(int c) { return cdiv(NCT(c), WC); }                     // column tiles per wave
    static constexpr int DYMIN(int c) { return dmin(cy(c)) + KY0(c); }
    static constexpr int DXMIN(int c) { return dmin(cx(c)); }
    static constexpr int LR(int c) { return OH + NTY(c) - 1; }                       // plane rows / columns held in LDS (ring included)
    static constexpr int LC(int c) { return OW + NTX(c) - 1; }
    static constexpr int PH(int c) { return AH > cy(c) ? (AH - 1 - cy(c)) / ST + 1 : 0; }     // rows / columns the plane really has
    static constexpr int PW(int c) { return AW > cx(c) ? (AW - 1 - cx(c)) / ST + 1 : 0; }
    static constexpr int IMGCELLS(int c) { return LR(c) * LC(c); }
    static constexpr int NCELL(int c) { return IB * IMGCELLS(c); }
    static constexpr int NCELLP(int c) { return rup(NCELL(c), 16); }
    // k-th tap of the class (ky-major) -> tap of the layer, and its cell offset against the row's base cell
    static constexpr int tap_ty(int c, int k) { return t0(cy(c)) + (KY0(c) + k / NTX(c)) * ST; }
    static constexpr int tap_tx(int c, int k) { return t0(cx(c)) + (k % NTX(c)) * ST; }
    static constexpr int tapcell(int c, int k) { return (k / NTX(c)) * LC(c) + k % NTX(c); }
    // base cell of pixel row kr of the batch (rows past the batch: cell 0 -- their small-side rows are zero)
    static constexpr int rowcell(int c, int kr) {
        if (kr >= ROWS) return 0;
        const int ib = kr / OYX, pix = kr % OYX;
        return ib * IMGCELLS(c) + (pix / OW) * LC(c) + pix % OW;
    }
    // ---- slot layout
    // (the small region is padded to a whole number of DMA rounds -- WAVES instructions of 1 KB -- so that round j of a fill
    //  belongs to ONE tensor for every wave: the descriptor of an instruction is a compile-time choice)
    static constexpr int SM_USED = NTN * RPAD * 64;
    static constexpr int SM_BYTES = rup(SM_USED, WAVES * 1024);
    static constexpr int SM_CHUNKS = SM_BYTES / 16;
    static constexpr int NFS = SM_BYTES / (WAVES * 1024);                            // DMA rounds of the small region
    static constexpr int BG_BYTES(int c) { return CT * NCELLP(c) * 64; }
    static constexpr int slot_bytes_of(int c) { return SM_BYTES + rup(BG_BYTES(c), WAVES * 1024); }
    static constexpr int max_slot() { int r = 0; for (int c = 0; c < NCLS; ++c) r = cmax(r, slot_bytes_of(c)); return r; }
    static constexpr int SLOT_BYTES = max_slot();
    static constexpr int NF = SLOT_BYTES / (WAVES * 1024);                           // DMA instructions per wave per fill (all classes alike)
    static constexpr int TOTAL = SLOTS * SLOT_BYTES;
    static constexpr int max_cpw() { int r = 0; for (int c = 0; c < NCLS; ++c) r = cmax(r, CPW(c)); return r; }
    // byte offsets inside a slot
    static constexpr int sm_off(int nt, int kr) { return (nt * RPAD + kr) * 64; }
    static constexpr int bg_off(int c, int ct, int cell) { return SM_BYTES + (ct * NCELLP(c) + cell) * 64; }

    // source of 16-byte chunk `ch` of the small region / chunk `ch2` of class c's big region
    static constexpr Src src_small(int ch) {
        const int nt = ch / (RPAD * 4), rem = ch % (RPAD * 4), kr = rem / 4, j = rem % 4;
        if (nt >= NTN || kr >= ROWS) return Src{-1, 0};
        return Src{0, (kr * N + nt * 32) * 2 + j * 16};
    }
    static constexpr Src src_big(int c, int ch2) {
        const int ct = ch2 / (NCELLP(c) * 4), rem = ch2 % (NCELLP(c) * 4), cell = rem / 4, j = rem % 4;
        if (ct >= CT || cell >= NCELL(c)) return Src{-1, 0};
        const int ib = cell / IMGCELLS(c), cc = cell % IMGCELLS(c);
        const int py = cc / LC(c) + DYMIN(c), px = cc % LC(c) + DXMIN(c);
        if (py < 0 || py >= PH(c) || px < 0 || px >= PW(c)) return Src{-1, 0};
        return Src{1, (((ib * AH + py * ST + cy(c)) * AW + px * ST + cx(c)) * C + ct * 32) * 2 + j * 16};
    }
    // source of 16-byte chunk `ch` of a class-c slot
    static constexpr Src src(int c, int ch) { return ch < SM_CHUNKS ? src_small(ch) : src_big(c, ch - SM_CHUNKS); }

    // ---- class pairs (PAIR): pair p = (the p-th heaviest class, the p-th lightest); slot = [small][big of A][big of B] in 1 KB pieces
    static constexpr int NPAIR = NCLS / 2;
    static constexpr int pair_a(int p) { return NCLS - 1 - p; }
    static constexpr int pair_b(int p) { return p; }
    static constexpr int p_bga(int) { return SM_USED; }
    static constexpr int p_bgb(int p) { return SM_USED + BG_BYTES(pair_a(p)); }
    static constexpr int p_bytes(int p) { return SM_USED + BG_BYTES(pair_a(p)) + BG_BYTES(pair_b(p)); }
    static constexpr int p_max() { int r = 0; for (int p = 0; p < NPAIR; ++p) r = cmax(r, p_bytes(p)); return r; }
    static constexpr int P_SLOT = rup(p_max(), 1024);
    static constexpr int P_TOTAL = SLOTS * P_SLOT;
    static constexpr int P_NF = cdiv(P_SLOT / 1024, 2 * WAVES);                      // DMA instructions per wave per fill (2*WAVES waves)
    static constexpr Src src_pair(int p, int ch) {
        if (ch < SM_USED / 16) return src_small(ch);
        if (ch < p_bgb(p) / 16) return src_big(pair_a(p), ch - SM_USED / 16);
        if (ch < p_bytes(p) / 16) return src_big(pair_b(p), ch - p_bgb(p) / 16);
        return Src{-1, 0};
    }
};

}  // namespace wrgeo
