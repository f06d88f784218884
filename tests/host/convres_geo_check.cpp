// Host check of multimodal-vae_amd/csrc/convres_geo.h: for every compiled geometry, every class row and every tap, the LDS
// address the kernel forms (lane base + tap immediate) must be the staged cell of the gathered pixel of gemm.h's gather
// semantics, or a cell no staged pixel ever occupies (the zero ring) when that pixel lies outside the image.
// Also checks the plan_fwdform / plan_classform values the launcher compares against (layers.h), restated here.
#include "convres_geo.h"
#include <cstdio>
#include <set>
#include <vector>

using namespace crgeo;
static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

template <class G>
void check(const char* name) {
    // staged cells
    std::set<int> staged;
    for (int iy = 0; iy < G::AH; ++iy)
        for (int ix = 0; ix < G::AW; ++ix) {
            const int c = G::cell(iy, ix);
            if (c < 0) continue;
            CHECK(c % 16 == 0 && c + G::PIX <= G::IMG_BYTES, "%s: cell(%d,%d)=%d out of image bytes %d", name, iy, ix, c, G::IMG_BYTES);
            CHECK(staged.insert(c).second, "%s: cell(%d,%d)=%d staged twice", name, iy, ix, c);
        }
    long long macs = 0;
    for (int c = 0; c < G::NCLS; ++c) {
        CHECK(G::K(c) % 16 == 0, "%s: K", name);
        for (int jy = 0; jy < G::OY(c); ++jy)
            for (int jx = 0; jx < G::OX(c); ++jx) {
                const int oy = G::out_y(c, jy), ox = G::out_x(c, jx);
                CHECK(oy >= 0 && oy < G::OH && ox >= 0 && ox < G::OW, "%s: class %d row (%d,%d) -> out (%d,%d)", name, c, jy, jx, oy, ox);
                const int base = G::base0(c) + jy * G::row_stride(c) + jx * G::col_stride(c);
                for (int ty = 0; ty < G::TH(c); ++ty)
                    for (int tx = 0; tx < G::TW(c); ++tx) {
                        const int iy = G::gy(c, jy, ty), ix = G::gx(c, jx, tx);
                        const int addr = base + G::tap_off(c, ty, tx);
                        CHECK(G::tap_off(c, ty, tx) >= 0 && G::tap_off(c, ty, tx) < 65536 - 512, "%s: tap immediate %d", name, G::tap_off(c, ty, tx));
                        CHECK(addr >= 0 && addr + G::PIX <= G::IMG_BYTES, "%s: class %d row (%d,%d) tap (%d,%d): addr %d outside image", name, c, jy, jx, ty, tx, addr);
                        const bool inside = iy >= 0 && iy < G::AH && ix >= 0 && ix < G::AW;
                        if (inside) {
                            CHECK(G::cell(iy, ix) == addr, "%s: class %d row (%d,%d) tap (%d,%d): pixel (%d,%d) cell %d != addr %d", name, c, jy, jx, ty, tx, iy, ix, G::cell(iy, ix), addr);
                            ++macs;
                        } else {
                            CHECK(staged.count(addr) == 0, "%s: class %d row (%d,%d) tap (%d,%d): outside pixel (%d,%d) hits staged cell %d", name, c, jy, jx, ty, tx, iy, ix, addr);
                        }
                        // the k-step immediates of this tap
                        const int tap = ty * G::TW(c) + tx;
                        for (int s = 0; s < G::KSTEP_PER_TAP; ++s) {
                            const int kk = tap * G::KSTEP_PER_TAP + s;
                            CHECK(G::step_off(c, kk) == G::tap_off(c, ty, tx) + s * 32, "%s: step_off", name);
                        }
                    }
            }
    }
    // every output pixel is produced by exactly one class row
    std::vector<int> seen(G::OH * G::OW, 0);
    for (int c = 0; c < G::NCLS; ++c)
        for (int jy = 0; jy < G::OY(c); ++jy)
            for (int jx = 0; jx < G::OX(c); ++jx) ++seen[G::out_y(c, jy) * G::OW + G::out_x(c, jx)];
    for (int v : seen) CHECK(v == 1, "%s: output pixel covered %d times", name, v);
    // the work equals the direct definition: pairs (output pixel of the big/small side, kernel tap) with an input inside
    long long direct = 0;
    if (G::FORM == 0) {       // out[oy][ox] += in[oy*S-P+kh][ox*S-P+kw]
        for (int oy = 0; oy < G::OH; ++oy) for (int ox = 0; ox < G::OW; ++ox)
            for (int kh = 0; kh < G::KH; ++kh) for (int kw = 0; kw < G::KW; ++kw) {
                const int iy = oy * G::S - G::PAD + kh, ix = ox * G::S - G::PAD + kw;
                if (iy >= 0 && iy < G::AH && ix >= 0 && ix < G::AW) ++direct;
            }
    } else {                  // big[oy][ox] += small[iy][ix] with oy = iy*S - P + kh
        for (int iy = 0; iy < G::AH; ++iy) for (int ix = 0; ix < G::AW; ++ix)
            for (int kh = 0; kh < G::KH; ++kh) for (int kw = 0; kw < G::KW; ++kw) {
                const int oy = iy * G::S - G::PAD + kh, ox = ix * G::S - G::PAD + kw;
                if (oy >= 0 && oy < G::OH && ox >= 0 && ox < G::OW) ++direct;
            }
    }
    CHECK(macs == direct, "%s: %lld in-image (row,tap) pairs, the convolution has %lld", name, macs, direct);
    // bank-slot rule behind the pixel pitch: 16 consecutive pixels -> 16 distinct 16-byte slots of the 256-byte bank row
    std::set<int> slots;
    for (int i = 0; i < 16; ++i) slots.insert((i * G::PIX / 16) % 16);
    CHECK(slots.size() == 16, "%s: pixel pitch %d is not an odd multiple of 16 bytes", name, G::PIX);
    printf("%-14s form %d  classes %d  image %6d B  in-image taps %lld\n", name, G::FORM, G::NCLS, G::IMG_BYTES, macs);
}

int main() {
    check<Geo<0, 32, 64, 25, 25, 12, 12, 4, 4, 2, 1>>("mm_conv2");
    check<Geo<0, 64, 128, 12, 12, 6, 6, 4, 4, 2, 1>>("mm_conv3");
    check<Geo<1, 128, 64, 6, 6, 12, 12, 4, 4, 2, 1>>("mm_convT2");
    check<Geo<1, 64, 32, 12, 12, 25, 25, 5, 5, 2, 1>>("mm_convT3");
    check<Geo<1, 64, 32, 12, 12, 25, 25, 4, 4, 2, 1>>("mm_conv2d");
    check<Geo<0, 32, 64, 25, 25, 12, 12, 5, 5, 2, 1>>("mm_convT3d");
    // CelebA / COCO shapes
    check<Geo<0, 32, 64, 32, 32, 16, 16, 4, 4, 2, 1>>("ca_conv2");
    check<Geo<0, 64, 128, 16, 16, 8, 8, 4, 4, 2, 1>>("ca_conv3");
    check<Geo<1, 128, 64, 8, 8, 16, 16, 4, 4, 2, 1>>("ca_convT2");
    check<Geo<1, 64, 32, 16, 16, 32, 32, 4, 4, 2, 1>>("ca_convT3");
    check<Geo<0, 128, 256, 8, 8, 5, 5, 4, 4, 1, 0>>("ca_conv4_s1");
    check<Geo<1, 256, 128, 5, 5, 8, 8, 4, 4, 1, 0>>("ca_convT1_s1");
    check<Geo<1, 256, 128, 4, 4, 8, 8, 4, 4, 2, 1>>("co_convT2");
    if (fails) { printf("%d failures\n", fails); return 1; }
    printf("ok\n");
    return 0;
}
