// Host-side helpers shared by the per-dataset model plans: geometry builders for the two gather forms,
// matching weight-pack descriptors, and a bump allocator over the caller-owned workspace.
#pragma once
#include "gemm.h"
#include "elementwise.h"
#include <vector>
#include <string>

struct ParamInfo {
    std::string name;
    int ndim;
    int shape[4];
    long long offset;   // element offset in the flat fp32 parameter / gradient buffers
    long long numel;
};

struct ConvGeom {       // a Conv2d (transposed=false) or ConvTranspose2d (transposed=true) layer
    int Cin, Cout, KH, KW, stride, pad;
    int IH, IW, OH, OW; // input / output spatial size of the layer's FORWARD
    bool transposed;
};

// Gather spec of one GEMM over pixels.  `fwdform`: rows are pixels of the SMALL (strided-to) side and taps walk
// the big side (Conv2d forward, ConvTranspose2d dgrad).  `classform`: rows are pixels of the BIG side split into
// stride-parity classes (ConvTranspose2d forward, Conv2d dgrad).
struct GatherPlan {
    GatherCommon c;
    GatherClass cls[MMVAE_MAX_CLASSES];
};

// big side [BH][BW], small side [SH][SW]; gathered tensor has Cg channels
inline GatherPlan plan_fwdform(int BH, int BW, int SH, int SW, int Cg, int KH, int KW, int stride, int pad, int N,
                               int groups, int group_n) {
    GatherPlan p{};
    p.c.groups = groups; p.c.group_n = group_n;
    p.c.AH = BH; p.c.AW = BW; p.c.Ald = Cg; p.c.C = Cg;
    p.c.sy = stride; p.c.sx = stride; p.c.dy = 1; p.c.dx = 1;
    p.c.OH = SH; p.c.OW = SW; p.c.osy = 1; p.c.osx = 1;
    p.c.N = N; p.c.nclasses = 1;
    GatherClass& k = p.cls[0];
    k.OY = SH; k.OX = SW; k.rows_per_group = group_n * SH * SW;
    k.TH = KH; k.TW = KW; k.offy = -pad; k.offx = -pad; k.ooy = 0; k.oox = 0;
    k.K = KH * KW * Cg; k.Kpad = round_up(k.K, 64);
    return p;
}

inline GatherPlan plan_classform(int BH, int BW, int SH, int SW, int Cg, int KH, int KW, int stride, int pad, int N,
                                 int groups, int group_n) {
    GatherPlan p{};
    p.c.groups = groups; p.c.group_n = group_n;
    p.c.AH = SH; p.c.AW = SW; p.c.Ald = Cg; p.c.C = Cg;
    p.c.sy = 1; p.c.sx = 1; p.c.dy = -1; p.c.dx = -1;
    p.c.OH = BH; p.c.OW = BW; p.c.osy = stride; p.c.osx = stride;
    p.c.N = N; p.c.nclasses = stride * stride;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            GatherClass& k = p.cls[ph * stride + pw];
            const int kh0 = (ph + pad) % stride, kw0 = (pw + pad) % stride;
            k.OY = (BH - ph + stride - 1) / stride; k.OX = (BW - pw + stride - 1) / stride;
            k.rows_per_group = group_n * k.OY * k.OX;
            k.TH = (KH - kh0 + stride - 1) / stride; k.TW = (KW - kw0 + stride - 1) / stride;
            k.offy = (ph + pad - kh0) / stride; k.offx = (pw + pad - kw0) / stride;
            k.ooy = ph; k.oox = pw;
            k.K = k.TH * k.TW * Cg; k.Kpad = round_up(k.K, 64);
        }
    return p;
}

inline GatherPlan plan_dense(int rows, int K, int ld, int N) {
    GatherPlan p = plan_fwdform(1, 1, 1, 1, K, 1, 1, 1, 0, N, 1, rows);
    p.c.Ald = ld;
    return p;
}

struct PackList {
    std::vector<PackDesc> d;
    long long mat_elems = 0, vec_elems = 0;
    // returns index; the packed matrix is [Npad][Kpad] at element offset d[i].dst_off
    int add(PackDesc x) {
        // one 256-thread block packs 256 vectors of 8 columns (elementwise.hip pack_kernel)
        x.first_block = d.empty() ? 0 : d.back().first_block + (int)(((long long)d.back().Npad * (d.back().Kpad / 8) + 255) / 256);
        if (x.is_f32) { x.dst_off = vec_elems; vec_elems += round_up(x.Npad * x.Kpad, 64); }
        else { x.dst_off = mat_elems; mat_elems += (long long)x.Npad * x.Kpad; }
        d.push_back(x);
        return (int)d.size() - 1;
    }
};

inline int npad_for(int N) {          // matches launch_gemm_gather's BN choice
    if (N <= 16) return 16;
    if (N <= 32) return 32;
    if (N <= 64) return 64;
    return round_up(N, 128);
}

// dense [N][K] matrix out of a row-major (N, Ksrc) parameter (optionally its transpose)
inline PackDesc pack_dense(long long src_off, int N, int K, int Npad, int Kpad, int s_n, int s_k) {
    PackDesc x{};
    x.src_off = src_off; x.N = N; x.K = K; x.Npad = Npad; x.Kpad = Kpad;
    x.NL = N > 0 ? N : 1; x.s_nhi = 0; x.s_nlo = s_n;
    x.TW = 1; x.C = K > 0 ? K : 1; x.s_ty = 0; x.s_tx = 0; x.s_c = s_k; x.o_ty = 0; x.o_tx = 0; x.step_t = 1;
    x.bias_off = -1;
    return x;
}

// weights for one class of a conv-like GEMM: n-stride / c-stride pick the tensor roles (see layers.h header)
inline PackDesc pack_conv(long long src_off, const GatherCommon& c, const GatherClass& k, int Npad, int s_n, int s_c, int KW,
                          int o_ty, int o_tx, int step) {
    PackDesc x{};
    x.src_off = src_off; x.N = c.N; x.K = k.K; x.Npad = Npad; x.Kpad = k.Kpad;
    x.NL = c.N; x.s_nhi = 0; x.s_nlo = s_n;
    x.TW = k.TW; x.C = c.C; x.s_ty = KW; x.s_tx = 1; x.s_c = s_c; x.o_ty = o_ty; x.o_tx = o_tx; x.step_t = step;
    x.bias_off = -1;
    return x;
}

class Workspace {
public:
    Workspace(void* base, size_t bytes) : base_((char*)base), cap_(bytes) {}
    template <typename T> T* take(size_t n) {
        size_t b = (n * sizeof(T) + 255) / 256 * 256;
        char* p = base_ ? base_ + off_ : nullptr;
        off_ += b;
        return reinterpret_cast<T*>(p);
    }
    size_t used() const { return off_; }
    bool ok() const { return off_ <= cap_; }
private:
    char* base_; size_t cap_; size_t off_ = 0;
};
