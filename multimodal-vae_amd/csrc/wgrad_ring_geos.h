// The layer geometries the ring-staged weight-gradient kernel (wgrad_ring.hip) is compiled for -- one list for the kernel's
// launcher and for the host-side address check (tests/host/wgrad_geo_check.cpp).
//   X(name,        C   N   AH  AW  OH  OW KH KW ST PAD NS IB WAVES SLOTS [WGQ [TGN [PAIR]]])
// C / AH / AW: channels and size of the BIG side, N / OH / OW: of the SMALL side (forward-form geometry, wgrad_geo.h);
// NS: small-side channels per workgroup, IB: images per ring slot, SLOTS: ring depth.
#pragma once
#define WGRAD_RING_GEOS(X)                                                                                                   \
    X(mm_convT3p,   32, 64, 25, 25, 12, 12, 5, 5, 2, 1, 64, 1, 4, 2, 0, 1, 1)   /* pair form, opt-in (knob wr_pair): two parity classes per 8-wave workgroup behind one fill of the small image */ \
    X(mm_convT3,    32, 64, 25, 25, 12, 12, 5, 5, 2, 1, 64, 1, 4, 2)   /* MultiMNIST hallucinate.6 (big side = output gradient) */ \
    X(mm_conv2,     32, 64, 25, 25, 12, 12, 4, 4, 2, 1, 64, 1, 4, 2)   /* features.2 (big side = layer input) */                 \
    X(mm_conv3,     64, 128, 12, 12, 6, 6, 4, 4, 2, 1, 64, 2, 4, 2)    /* features.5 and hallucinate.3 */                      \
    X(mm_conv4,     128, 256, 6, 6, 2, 2, 4, 4, 2, 0, 64, 16, 4, 2, 1) /* features.8 and hallucinate.0: 2x2 <-> 6x6, 2 MB of dW, 16 images per batch */ \
    X(ca_conv2,     32, 64, 32, 32, 16, 16, 4, 4, 2, 1, 64, 1, 4, 2)   /* CelebA features.2 and hallucinate.6 (celeba/model.py:104,146) */ \
    X(ca_conv3,     64, 128, 16, 16, 8, 8, 4, 4, 2, 1, 64, 2, 4, 2)    /* CelebA features.5 and hallucinate.3 */               \
    X(ca_conv4,     128, 256, 8, 8, 5, 5, 4, 4, 1, 0, 64, 4, 4, 2, 4, 4) /* CelebA features.8 and hallucinate.0: stride 1, 8x8 <-> 5x5, 2 MB of dW, one tap row per class */
