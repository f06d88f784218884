// Host check of multimodal-vae_amd/csrc/wgrad_geo.h (the ring-staged weight-gradient kernel, wgrad_ring.hip).
// For every compiled geometry and every stride-parity class it replays on the CPU exactly what the kernel does with addresses:
//   1. the DMA fill: every 16-byte chunk of a slot is written from Geo::src (or zero-filled), from tensors whose elements are
//      their own coordinates (image, y, x, channel);
//   2. the fragment reads: for every pixel row of the batch, every tap of the class and every channel tile, the small-side
//      granule at sm_off and the big-side granule at bg_off(rowcell + tapcell);
// and compares what those reads return with the definition of the weight gradient
//      dW[n][(ty,tx)][c] += S[img][oy][ox][n] * B[img][oy*ST-PAD+ty][ox*ST-PAD+tx][c]      (zero outside the image).
// It also checks that the classes partition the taps, that the copies' column order maps back to the layer's taps the way the
// reduce kernel assumes, and the DMA-round / bank-layout assumptions of the kernel.
#include "wgrad_geo.h"
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

struct Elem { short img, y, x, ch; };           // a bf16 element stands for its coordinates; {-1,..} = zero
static const Elem ZERO{-1, 0, 0, 0};
static bool same(const Elem& a, const Elem& b) { return a.img == b.img && (a.img < 0 || (a.y == b.y && a.x == b.x && a.ch == b.ch)); }

template <class G>
void check(const char* name) {
    static_assert(G::SLOT_BYTES % (G::WAVES * 1024) == 0 && G::SM_BYTES % (G::WAVES * 1024) == 0, "DMA rounds");
    static_assert(G::TOTAL <= 160 * 1024, "LDS budget");
    // taps are partitioned by the classes; the class-local order (ky-major) is the layer's order restricted to the class
    std::set<int> seen;
    for (int c = 0; c < G::NCLS; ++c) {
        CHECK(G::ntap(G::cy(c), G::KH) % G::TGN == 0, "%s: %d tap rows of a parity class do not split into %d groups", name, G::ntap(G::cy(c), G::KH), G::TGN);
        int prev = -1;
        for (int k = 0; k < G::NTAPS(c); ++k) {
            const int ty = G::tap_ty(c, k), tx = G::tap_tx(c, k);
            CHECK(ty >= 0 && ty < G::KH && tx >= 0 && tx < G::KW, "%s: class %d tap %d -> (%d,%d)", name, c, k, ty, tx);
            CHECK(G::par(ty) == G::cy(c) && G::par(tx) == G::cx(c), "%s: class %d tap %d parity", name, c, k);
            CHECK(seen.insert(ty * G::KW + tx).second, "%s: tap (%d,%d) in two classes", name, ty, tx);
            CHECK(ty * G::KW + tx > prev, "%s: class order", name);
            prev = ty * G::KW + tx;
            // the reduce kernel's formula
            const int ntx = G::NTX(c);
            CHECK((G::tap_ty(c, 0) + (k / ntx) * G::ST) == ty && (G::tap_tx(c, 0) + (k % ntx) * G::ST) == tx, "%s: reduce tap formula", name);
        }
        CHECK(G::slot_bytes_of(c) <= G::SLOT_BYTES, "%s: slot", name);
        CHECK(G::bg_off(c, G::CT - 1, G::NCELLP(c) - 1) + 64 <= G::SLOT_BYTES, "%s: big region overruns the slot", name);
    }
    CHECK((int)seen.size() == G::KH * G::KW, "%s: %d of %d taps covered", name, (int)seen.size(), G::KH * G::KW);

    long long macs = 0, direct = 0;
    for (int c = 0; c < G::NCLS; ++c) {
        // ---- 1. DMA fill of a slot (batch = images 0 .. IB-1)
        std::vector<Elem> lds(G::SLOT_BYTES / 2, Elem{-2, 0, 0, 0});        // -2: never written
        for (int ch = 0; ch < G::SLOT_BYTES / 16; ++ch) {
            const wrgeo::Src s = G::src(c, ch);
            for (int e = 0; e < 8; ++e) {
                Elem v = ZERO;
                if (s.tensor == 0) {
                    const int el = s.off / 2 + e;                              // element of S[img][pix][n] (slice offset 0)
                    const int n = el % G::N, row = el / G::N;
                    CHECK(s.off % 16 == 0 && row < G::ROWS, "%s: small source row %d", name, row);
                    v = Elem{(short)(row / G::OYX), (short)((row % G::OYX) / G::OW), (short)((row % G::OYX) % G::OW), (short)n};
                } else if (s.tensor == 1) {
                    const int el = s.off / 2 + e;                              // element of B[img][y][x][c]
                    const int chn = el % G::C, px = el / G::C;
                    const int x = px % G::AW, y = (px / G::AW) % G::AH, img = px / (G::AW * G::AH);
                    CHECK(s.off % 16 == 0 && img < G::IB, "%s: big source image %d", name, img);
                    v = Elem{(short)(100 + img), (short)y, (short)x, (short)chn};
                }
                lds[ch * 8 + e] = v;
            }
        }
        // ---- 2. fragment reads of every (row, tap, tiles)
        for (int kr = 0; kr < G::RPAD; ++kr) {
            for (int nt = 0; nt < G::NTN; ++nt)
                for (int e = 0; e < 32; ++e) {
                    const Elem got = lds[G::sm_off(nt, kr) / 2 + e];
                    Elem want = ZERO;
                    if (kr < G::ROWS) want = Elem{(short)(kr / G::OYX), (short)((kr % G::OYX) / G::OW), (short)((kr % G::OYX) % G::OW), (short)(nt * 32 + e)};
                    CHECK(same(got, want), "%s: class %d small row %d tile %d elem %d: got img %d", name, c, kr, nt, e, got.img);
                }
            if (kr >= G::ROWS) {
                CHECK(G::rowcell(c, kr) == 0, "%s: padding row cell", name);
                continue;
            }
            const int img = kr / G::OYX, oy = (kr % G::OYX) / G::OW, ox = (kr % G::OYX) % G::OW;
            for (int k = 0; k < G::NTAPS(c); ++k) {
                const int ty = G::tap_ty(c, k), tx = G::tap_tx(c, k);
                const int iy = oy * G::ST - G::PAD + ty, ix = ox * G::ST - G::PAD + tx;
                const bool inside = iy >= 0 && iy < G::AH && ix >= 0 && ix < G::AW;
                const int cell = G::rowcell(c, kr) + G::tapcell(c, k);
                CHECK(cell >= 0 && cell < G::NCELL(c), "%s: class %d row %d tap %d: cell %d of %d", name, c, kr, k, cell, G::NCELL(c));
                for (int ct = 0; ct < G::CT; ++ct)
                    for (int e = 0; e < 32; ++e) {
                        const Elem got = lds[G::bg_off(c, ct, cell) / 2 + e];
                        const Elem want = inside ? Elem{(short)(100 + img), (short)iy, (short)ix, (short)(ct * 32 + e)} : ZERO;
                        CHECK(same(got, want), "%s: class %d row %d (img %d, %d,%d) tap (%d,%d) tile %d elem %d: got (%d,%d,%d,%d) want (%d,%d,%d,%d)", name, c,
                              kr, img, oy, ox, ty, tx, ct, e, got.img, got.y, got.x, got.ch, want.img, want.y, want.x, want.ch);
                    }
                if (inside) ++macs;
            }
        }
    }
    for (int img = 0; img < G::IB; ++img)
        for (int oy = 0; oy < G::OH; ++oy) for (int ox = 0; ox < G::OW; ++ox)
            for (int ty = 0; ty < G::KH; ++ty) for (int tx = 0; tx < G::KW; ++tx) {
                const int iy = oy * G::ST - G::PAD + ty, ix = ox * G::ST - G::PAD + tx;
                if (iy >= 0 && iy < G::AH && ix >= 0 && ix < G::AW) ++direct;
            }
    CHECK(macs == direct, "%s: %lld in-image (row, tap) pairs, the convolution has %lld", name, macs, direct);
    // four consecutive granules = one 256-byte bank row: granule addresses are multiples of 64
    CHECK(G::SM_BYTES % 64 == 0, "%s: granules", name);
    printf("%-14s classes %d  slot %6d B x %d  k-steps %2d  column tiles/wave %d  DMA rounds %d (%d small)  %s\n", name, G::NCLS, G::SLOT_BYTES, G::SLOTS,
           G::KST, G::max_cpw(), G::NF, G::NFS, fails ? "FAIL" : "ok");
}


// Pair form (Geo::PAIR): one slot = [small][big of class A][big of class B], filled by src_pair in 1 KB pieces; the waves of class X
// read their big region at its base p_bga / p_bgb with the class's own cell layout.
template <class G>
void check_pair(const char* name) {
    if constexpr (G::PAIR != 0) {
        static_assert(G::NCLS % 2 == 0 && G::SLOTS == 2, "pairs of classes, two slots");
        static_assert(G::P_TOTAL <= 160 * 1024, "LDS budget");
        static_assert(G::SM_USED % 1024 == 0, "the small region ends on a DMA piece");
        std::set<int> covered;
        for (int p = 0; p < G::NPAIR; ++p) {
            const int cls[2] = {G::pair_a(p), G::pair_b(p)}, base[2] = {G::p_bga(p), G::p_bgb(p)};
            CHECK(covered.insert(cls[0]).second && covered.insert(cls[1]).second, "%s: pair %d repeats a class", name, p);
            CHECK(G::BG_BYTES(cls[0]) % 1024 == 0 && G::BG_BYTES(cls[1]) % 1024 == 0, "%s: pair %d: a DMA piece straddles two regions", name, p);
            CHECK(G::p_bytes(p) <= G::P_SLOT && G::P_NF * 2 * G::WAVES * 1024 >= G::p_bytes(p), "%s: pair %d: pieces", name, p);
            std::vector<Elem> lds(G::P_SLOT / 2, Elem{-2, 0, 0, 0});
            for (int ch = 0; ch < G::P_SLOT / 16; ++ch) {
                const wrgeo::Src s = G::src_pair(p, ch);
                // (the kernel picks the descriptor per 1 KB piece: pieces below SM_USED are small-side)
                CHECK(s.tensor < 0 || (s.tensor == 0) == (ch * 16 < G::SM_USED), "%s: pair %d chunk %d: tensor %d on the wrong side of SM_USED", name, p, ch, s.tensor);
                for (int e = 0; e < 8; ++e) {
                    Elem v = ZERO;
                    if (s.tensor == 0) {
                        const int el = s.off / 2 + e, n = el % G::N, row = el / G::N;
                        CHECK(row < G::ROWS, "%s: small source row %d", name, row);
                        v = Elem{(short)(row / G::OYX), (short)((row % G::OYX) / G::OW), (short)((row % G::OYX) % G::OW), (short)n};
                    } else if (s.tensor == 1) {
                        const int el = s.off / 2 + e, chn = el % G::C, px = el / G::C;
                        const int x = px % G::AW, y = (px / G::AW) % G::AH, img = px / (G::AW * G::AH);
                        CHECK(img < G::IB, "%s: big source image %d", name, img);
                        v = Elem{(short)(100 + img), (short)y, (short)x, (short)chn};
                    }
                    lds[ch * 8 + e] = v;
                }
            }
            for (int kr = 0; kr < G::RPAD; ++kr) {
                for (int nt = 0; nt < G::NTN; ++nt)
                    for (int e = 0; e < 32; ++e) {
                        const Elem got = lds[G::sm_off(nt, kr) / 2 + e];
                        Elem want = ZERO;
                        if (kr < G::ROWS) want = Elem{(short)(kr / G::OYX), (short)((kr % G::OYX) / G::OW), (short)((kr % G::OYX) % G::OW), (short)(nt * 32 + e)};
                        CHECK(same(got, want), "%s: pair %d small row %d tile %d elem %d", name, p, kr, nt, e);
                    }
                if (kr >= G::ROWS) continue;
                const int img = kr / G::OYX, oy = (kr % G::OYX) / G::OW, ox = (kr % G::OYX) % G::OW;
                for (int side = 0; side < 2; ++side) {
                    const int c = cls[side];
                    for (int k = 0; k < G::NTAPS(c); ++k) {
                        const int ty = G::tap_ty(c, k), tx = G::tap_tx(c, k);
                        const int iy = oy * G::ST - G::PAD + ty, ix = ox * G::ST - G::PAD + tx;
                        const bool inside = iy >= 0 && iy < G::AH && ix >= 0 && ix < G::AW;
                        const int cell = G::rowcell(c, kr) + G::tapcell(c, k);
                        for (int ct = 0; ct < G::CT; ++ct)
                            for (int e = 0; e < 32; ++e) {
                                const Elem got = lds[(base[side] + (ct * G::NCELLP(c) + cell) * 64) / 2 + e];
                                const Elem want = inside ? Elem{(short)(100 + img), (short)iy, (short)ix, (short)(ct * 32 + e)} : ZERO;
                                CHECK(same(got, want), "%s: pair %d class %d row %d tap (%d,%d) tile %d elem %d: got (%d,%d,%d,%d)", name, p, c, kr, ty, tx, ct, e,
                                      got.img, got.y, got.x, got.ch);
                            }
                    }
                }
            }
        }
        CHECK((int)covered.size() == G::NCLS, "%s: the pairs cover %d of %d classes", name, (int)covered.size(), G::NCLS);
        printf("%-14s pair form: slot %6d B x %d  DMA pieces/wave %d  %s\n", name, G::P_SLOT, G::SLOTS, G::P_NF, fails ? "FAIL" : "ok");
    }
}

#include "wgrad_ring_geos.h"

int main() {
#define X(name, ...) check<wrgeo::Geo<__VA_ARGS__>>(#name); check_pair<wrgeo::Geo<__VA_ARGS__>>(#name);
    WGRAD_RING_GEOS(X)
#undef X
    if (fails) { printf("%d failures\n", fails); return 1; }
    printf("ok\n");
    return 0;
}
