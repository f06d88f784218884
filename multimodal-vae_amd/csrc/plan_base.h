// Shared host-side machinery of the per-dataset model plans (multimnist.hip, mnist.hip, celeba.hip): the flat
// parameter table, weight-pack descriptor lists, layer descriptors, GEMM / wgrad parameter builders with their
// split-K heuristics, the BatchNorm+activation launcher and the multi-stream fork/join helpers.
// Include only from .hip translation units (it defines a few tiny __global__ helpers).
#pragma once
#include "layers.h"
#include <map>
#include <mutex>
#include <functional>
#include <utility>
#include <cstdlib>

constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f, DROP_P = 0.1f;

struct ModelBuffers {       // caller-owned device memory bound to a plan
    float* params = nullptr;        // flat fp32 parameters, state_dict order
    float* grads = nullptr;         // flat fp32 gradients
    float* bn_stats = nullptr;      // running_mean | running_var of every BatchNorm, state_dict order
    long long* bn_nbt = nullptr;    // num_batches_tracked
    bf16* packed = nullptr;         // bf16 GEMM-layout weights
    float* packed_vec = nullptr;    // fp32 packed vectors
    float* gpk = nullptr;           // fp32 packed weight gradients
    float* gpk_vec = nullptr;
    PackDesc* desc_dev = nullptr;   // device copies of the descriptor tables
    PackDesc* gdesc_dev = nullptr;
};

struct ConvL {
    ConvGeom g;
    long long w_off;
    int bn;                       // index into bn tables or -1
    GatherPlan fwd, dgrad;        // geometry templates (groups/pointers filled per call)
    int pk_fwd[4], pk_dgrad[4], gk[4];
    int pk_fwd_f[4] = {-1, -1, -1, -1}, pk_dgrad_f[4] = {-1, -1, -1, -1};   // fragment-major copies (add_frag_packs) or -1
    bool wgrad_fwdform = false;   // transposed conv: weight gradient in the data-gradient geometry (gk[0] only)
};
struct LinL {
    long long w_off, b_off;
    int N, K;
    int pk_fwd, pk_dgrad, gk;
    int pk_dgrad4[4];             // classifier.0 only: one dgrad matrix per pixel of the 2x2 feature map
};
struct BnL { long long w_off, b_off; int C; long long stat_off; int idx; };


struct PlanBase {
    int D = 0, B = 0;
    std::vector<ParamInfo> params;
    std::map<std::string, int> pidx;
    long long nparams = 0;
    PackList pk, gk;
    ModelBuffers buf;
    bool bound = false;
    // side streams: independent branches of the step (second-modality path, weight gradients) run beside the main
    // chain; forks/joins are event edges, so nothing syncs the host (and a captured HIP graph gets parallel branches)
    hipStream_t st_text = nullptr, st_wgrad = nullptr, st_wgrad2 = nullptr, st_wgrad2_own = nullptr;
    int wgrad_rr = 0;               // weight gradients alternate between the two side streams
    bool defer_wgrad = false;       // collect weight-gradient launches instead of issuing them (flush_wgrads)
    std::vector<WgradParams> deferred;
    std::vector<hipEvent_t> events;
    size_t next_event = 0;
    hipEvent_t fork_ev = nullptr;   // the event of the last arm_fork / commit_fork pair
    // side-stream work waiting for the next fork (side_later / side_flush); lane 1: onto the second-modality stream (st_text) instead
    // of a weight-gradient stream -- the tail of the step, when that stream has run dry
    std::vector<std::pair<std::function<int(hipStream_t)>, int>> side_pending;
    bool in_step = false;           // a fused multi-stream step is being enqueued (arm_fork / commit_fork are no-ops otherwise)
    bool capturing = false;         // ... into a HIP graph: forks are plain event records (a kernel's completion event is not a capture node --
                                    // the side streams would stay outside the graph and their work would be missing from every replay)
    hipEvent_t ev_early = nullptr;  // data-parallel step: the early gradient part is complete in the flat buffer
    bool wgrad_forked = false;
    bool spare_used = false;        // side_flush put work on the spare weight-gradient stream (st_wgrad2_own): join_sides joins it
    bool batch_reduce = false;      // set by side_flush: the weight gradients it issues leave their partial copies to ONE reduce
                                    // launch per stream at the end of the flush (was: one behind every weight gradient)
    unsigned dec_skip_mask = 0;     // bit k: pass k is absent from this step (weak-supervision variants): its decoder BatchNorm
                                    // group leaves the running statistics alone
    bool no_pack = false;           // the plan never reads the packed bf16 weights (fp32 MNIST path): pack_weights is a no-op
    bool single_wgrad_stream = false;   // the plan's weight gradients all go to ONE side stream (MultiMNIST: 0.967 -> 0.950 ms
                                        // per step; CelebA is 3 % slower that way and keeps two)
    bool no_splitk = false;         // set while enqueueing on a side stream: the split-K slabs belong to the main chain
    WgradSlabCtx slab;              // weight-gradient partial-tile slabs of the running step (pool carved from the workspace)
    // split-K partial slabs (carved from the caller's workspace)
    float* sk_buf = nullptr; size_t sk_floats = 0; unsigned* sk_cnt = nullptr;
    // generic queries (capi.cpp): BatchNorm layers in state_dict order and the workspace size
    std::vector<std::string> bn_names;
    std::vector<BnL> bn_list;
    size_t ws_bytes = 0;
    // the granular module entry points run ONE pass (groups = variants = 1): their workspace is carved for B rows where the
    // fused step needs 3B / 2B (carve_passes = 1 instead of 3), mmvae_<family>_module_workspace_bytes
    size_t ws_bytes_module = 0;
    int carve_passes = 3;
    virtual ~PlanBase() {}
};


inline int edge(PlanBase& P, hipStream_t from, hipStream_t to);
inline hipEvent_t next_ev(PlanBase& P);

// weight gradients only feed the optimizer: when the step runs multi-stream they go to the side stream
inline int wgrad_async(PlanBase& P, const WgradParams& g, hipStream_t s) {
    const bool serial = mmvae_serial();
    if (!P.wgrad_forked || serial) return launch_wgrad(g, s, &P.slab);
    if (P.defer_wgrad) { P.deferred.push_back(g); return MMVAE_OK; }
    hipStream_t w = (P.wgrad_rr++ & 1) ? P.st_wgrad2 : P.st_wgrad;
    MMVAE_TRY(edge(P, s, w));
    MMVAE_TRY(launch_wgrad(g, w, &P.slab));
    // the slab copies are summed right behind the kernel on the SAME side stream: off the main chain (one reduce launch
    // at the end of the step would read every slab of the step on the critical tail)
    return launch_wgrad_reduce(&P.slab, w, true);
}

// ---- forks bound to a kernel's completion (common.h: MMVAE_LAUNCH)
//   arm_fork(P);  <launch ONE main-chain kernel>;  commit_fork(P, s);  then fork_to(P, stream) any number of times
// A launcher that does not go through MMVAE_LAUNCH leaves the event unclaimed: commit_fork records it the old way.
inline void arm_fork(PlanBase& P) {
    if (!P.in_step) return;
    P.fork_ev = next_ev(P);
    if (!P.capturing && !mmvae_knob("no_stop_events", 0)) mmvae_arm_stop_event(P.fork_ev);
}
inline int commit_fork(PlanBase& P, hipStream_t s) {
    if (!P.in_step) return MMVAE_OK;
    const bool unclaimed = mmvae_take_stop_event() != nullptr || P.capturing || mmvae_knob("no_stop_events", 0);
    if (unclaimed && hipEventRecord(P.fork_ev, s) != hipSuccess) {
        mmvae_set_error("stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}
inline int fork_to(PlanBase& P, hipStream_t to) {
    {
        const int k = mmvae_knob("dbg_skip_edges", 0);
        if (k == 1 || k == 2) return MMVAE_OK;
    }
    if (hipStreamWaitEvent(to, P.fork_ev, 0) != hipSuccess) {
        mmvae_set_error("stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}
// the weight-gradient side stream (alternating when the plan has two), or `s` when the step is not forked; the caller has
// committed a fork event for the kernel that produced the operands
inline int wgrad_fork(PlanBase& P, hipStream_t s, hipStream_t* w) {
    const bool serial = mmvae_serial();
    if (!P.wgrad_forked || serial) { *w = s; return MMVAE_OK; }
    *w = (P.wgrad_rr++ & 1) ? P.st_wgrad2 : P.st_wgrad;
    return fork_to(P, *w);
}

// Side-stream work (weight gradients and the elementwise passes in front of them) is collected and issued behind the NEXT
// fork the main chain makes anyway: every fork costs the main chain ~5-6 us, and the one weight-gradient stream carries a
// backlog through most of the backward pass, so a fork per layer bought nothing.
inline void side_later(PlanBase& P, std::function<int(hipStream_t)> fn, int lane = 0) { P.side_pending.push_back({std::move(fn), lane}); }
inline int side_flush(PlanBase& P, hipStream_t s) {
    if (P.side_pending.empty()) return MMVAE_OK;
    const bool serial = mmvae_serial();
    int rc = MMVAE_OK;
    P.batch_reduce = mmvae_knob("batch_reduce", 1) != 0;
    if (!P.wgrad_forked || serial) {
        for (auto& fn : P.side_pending)
            if (rc == MMVAE_OK) rc = fn.first(s);
        P.side_pending.clear();
        if (rc == MMVAE_OK && P.batch_reduce && !(mmvae_knob("mm_wgrad_inline", 0) == 2)) rc = launch_wgrad_reduce(&P.slab, s, true);
        P.batch_reduce = false;
        return rc;
    }
    // the pieces are independent of each other: they alternate between the weight-gradient streams (each of these kernels is a
    // latency chain on a fraction of the chip; two of them side by side finish sooner than one after the other)
    hipStream_t w[4] = {P.st_wgrad, P.st_wgrad2, P.st_text, P.st_wgrad2_own};
    bool forked[4] = {false, false, false, false};
    const int lanes = mmvae_knob("side_lanes", 3);          // bit 0: lane 1 -> the second-modality stream, bit 1: lane 2 -> the spare weight-gradient stream
    for (auto& fn : P.side_pending) {
        const int i = (fn.second == 1 && P.st_text && (lanes & 1)) ? 2 : (fn.second == 2 && P.st_wgrad2_own && (lanes & 2)) ? 3 :
                      (w[0] == w[1]) ? 0 : (P.wgrad_rr++ & 1);
        if (!forked[i]) { forked[i] = true; if (rc == MMVAE_OK) rc = fork_to(P, w[i]); }
        if (i == 3) P.spare_used = true;
        if (rc == MMVAE_OK) rc = fn.first(w[i]);
    }
    P.side_pending.clear();
    for (int i = 0; i < 4; ++i)
        if (forked[i] && P.batch_reduce && rc == MMVAE_OK) rc = launch_wgrad_reduce(&P.slab, w[i], true);
    P.batch_reduce = false;
    return rc;
}

// The side stream a weight gradient issued now would run on (after an edge from `s`), or `s` itself when the step is not forked:
// lets the caller put the elementwise pass that prepares the gradient's operand in front of it, off the main chain.
inline int wgrad_side_stream(PlanBase& P, hipStream_t s, hipStream_t* w) {
    const bool serial = mmvae_serial();
    if (!P.wgrad_forked || serial) { *w = s; return MMVAE_OK; }
    *w = (P.wgrad_rr++ & 1) ? P.st_wgrad2 : P.st_wgrad;
    return edge(P, s, *w);
}
inline int wgrad_on(PlanBase& P, const WgradParams& g, hipStream_t w) {
    if (mmvae_knob("wgrad_atomic", 0)) return launch_wgrad(g, w, nullptr);      // A/B: fp32 atomics into the packed gradient, no slab / reduce
    MMVAE_TRY(launch_wgrad(g, w, &P.slab));
    if (!P.wgrad_forked && mmvae_knob("mm_wgrad_inline", 0) == 2) return MMVAE_OK;      // summed at the end of the step, one launch
    if (P.batch_reduce) return MMVAE_OK;                                                 // summed at the end of the flush
    return launch_wgrad_reduce(&P.slab, w, true);
}

// issue the weight gradients collected while defer_wgrad was set (their operands are complete on `s` by now)
inline int flush_wgrads(PlanBase& P, hipStream_t s) {
    P.defer_wgrad = false;
    for (const WgradParams& g : P.deferred) {
        hipStream_t w = (P.wgrad_rr++ & 1) ? P.st_wgrad2 : P.st_wgrad;
        MMVAE_TRY(edge(P, s, w));
        MMVAE_TRY(launch_wgrad(g, w, &P.slab));
    }
    P.deferred.clear();
    return MMVAE_OK;
}

inline void add_param(PlanBase& P, const std::string& name, std::initializer_list<int> shape) {
    ParamInfo pi{};
    pi.name = name; pi.ndim = (int)shape.size(); pi.numel = 1;
    int i = 0;
    for (int s : shape) { pi.shape[i++] = s; pi.numel *= s; }
    pi.offset = P.nparams;
    P.nparams += pi.numel;
    P.pidx[name] = (int)P.params.size();
    P.params.push_back(pi);
}
inline long long off(const PlanBase& P, const std::string& n) { return P.params[P.pidx.at(n)].offset; }

// ---- conv-like layer: builds both gather plans and the pack descriptors
inline void build_conv(PlanBase& P, ConvL& L, const std::string& wname, ConvGeom g, int bn_idx, bool need_dgrad, bool thin_in, bool thin_out) {
    L.g = g; L.w_off = off(P, wname); L.bn = bn_idx;
    const int kk = g.KH * g.KW;
    for (int i = 0; i < 4; ++i) { L.pk_fwd[i] = L.pk_dgrad[i] = L.gk[i] = -1; }
    if (!g.transposed) {
        // Conv2d weight (Cout, Cin, KH, KW)
        if (thin_in) {      // dense over im2col patches: K = kk*Cin
            const int K = kk * g.Cin, ld = round_up(K, 8);
            L.fwd = plan_dense(1, ld, ld, g.Cout);
            PackDesc d = pack_dense(L.w_off, g.Cout, K, npad_for(g.Cout), L.fwd.cls[0].Kpad, g.Cin * kk, 1);
            // patches order k = (kh*KW+kw)*Cin + ci ; weight order ci*kk + kh*KW + kw
            d.TW = g.KW; d.C = g.Cin; d.s_ty = g.KW; d.s_tx = 1; d.s_c = kk;
            L.pk_fwd[0] = P.pk.add(d);
            PackDesc gd = d; gd.Npad = round_up(g.Cout, 64);
            L.gk[0] = P.gk.add(gd);
        } else {
            L.fwd = plan_fwdform(g.IH, g.IW, g.OH, g.OW, g.Cin, g.KH, g.KW, g.stride, g.pad, g.Cout, 1, 1);
            PackDesc d = pack_conv(L.w_off, L.fwd.c, L.fwd.cls[0], npad_for(g.Cout), g.Cin * kk, kk, g.KW, 0, 0, 1);
            L.pk_fwd[0] = P.pk.add(d);
            PackDesc gd = d; gd.Npad = round_up(g.Cout, 64);
            L.gk[0] = P.gk.add(gd);
        }
        if (need_dgrad) {   // rows over the input (big) side, gathers dr (small side, Cout channels)
            L.dgrad = plan_classform(g.IH, g.IW, g.OH, g.OW, g.Cout, g.KH, g.KW, g.stride, g.pad, g.Cin, 1, 1);
            for (int ci = 0; ci < L.dgrad.c.nclasses; ++ci) {
                const int ph = ci / g.stride, pw = ci % g.stride;
                PackDesc d = pack_conv(L.w_off, L.dgrad.c, L.dgrad.cls[ci], npad_for(g.Cin), kk, g.Cin * kk, g.KW,
                                       (ph + g.pad) % g.stride, (pw + g.pad) % g.stride, g.stride);
                L.pk_dgrad[ci] = P.pk.add(d);
            }
        }
    } else {
        // ConvTranspose2d weight (Cin, Cout, KH, KW); forward rows over the output (big) side
        if (!thin_out) {
            L.fwd = plan_classform(g.OH, g.OW, g.IH, g.IW, g.Cin, g.KH, g.KW, g.stride, g.pad, g.Cout, 1, 1);
            for (int ci = 0; ci < L.fwd.c.nclasses; ++ci) {
                const int ph = ci / g.stride, pw = ci % g.stride;
                PackDesc d = pack_conv(L.w_off, L.fwd.c, L.fwd.cls[ci], npad_for(g.Cout), kk, g.Cout * kk, g.KW,
                                       (ph + g.pad) % g.stride, (pw + g.pad) % g.stride, g.stride);
                L.pk_fwd[ci] = P.pk.add(d);
                if (!need_dgrad) {
                    PackDesc gd = d; gd.Npad = round_up(g.Cout, 64);
                    L.gk[ci] = P.gk.add(gd);
                }
            }
            if (need_dgrad) {
                L.dgrad = plan_fwdform(g.OH, g.OW, g.IH, g.IW, g.Cout, g.KH, g.KW, g.stride, g.pad, g.Cin, 1, 1);
                PackDesc d = pack_conv(L.w_off, L.dgrad.c, L.dgrad.cls[0], npad_for(g.Cin), g.Cout * kk, kk, g.KW, 0, 0, 1);
                L.pk_dgrad[0] = P.pk.add(d);
                // The weight gradient is taken in the SAME (forward-form) geometry: rows over the small input grid, the
                // plain operand is the layer input [rows][Cin], the gathered one the output gradient (all taps, K =
                // taps*Cout).  Against the class form (rows over the 4x larger output grid, one K = taps/4 * Cin per parity
                // class) it has 4x fewer rows, half the gathered bytes and twice the MFMAs per staged row.
                PackDesc gd = d; gd.Npad = round_up(g.Cin, 64);
                L.gk[0] = P.gk.add(gd);
                L.wgrad_fwdform = true;
            }
        } else {
            // thin output (Cout small): forward still class-form (N = Cout padded to 16); backward goes through
            // im2col patches of dlogit: dgrad = dense [rows][kk*Cout] x W_D[Cin][kk*Cout], wgrad in the same layout
            L.fwd = plan_classform(g.OH, g.OW, g.IH, g.IW, g.Cin, g.KH, g.KW, g.stride, g.pad, g.Cout, 1, 1);
            for (int ci = 0; ci < L.fwd.c.nclasses; ++ci) {
                const int ph = ci / g.stride, pw = ci % g.stride;
                PackDesc d = pack_conv(L.w_off, L.fwd.c, L.fwd.cls[ci], npad_for(g.Cout), kk, g.Cout * kk, g.KW,
                                       (ph + g.pad) % g.stride, (pw + g.pad) % g.stride, g.stride);
                L.pk_fwd[ci] = P.pk.add(d);
            }
            const int K = kk * g.Cout, ld = round_up(K, 8);
            L.dgrad = plan_dense(1, ld, ld, g.Cin);
            PackDesc d = pack_dense(L.w_off, g.Cin, K, npad_for(g.Cin), L.dgrad.cls[0].Kpad, g.Cout * kk, 1);
            d.TW = g.KW; d.C = g.Cout; d.s_ty = g.KW; d.s_tx = 1; d.s_c = kk;   // k = (kh*KW+kw)*Cout + co
            L.pk_dgrad[0] = P.pk.add(d);
            PackDesc gd = d; gd.Npad = round_up(g.Cin, 64);
            L.gk[0] = P.gk.add(gd);
        }
    }
}

// Fragment-major copies (PackDesc::frag) of a conv layer's forward / data-gradient weight packs: what the direct-B
// image-resident kernels stream (convres.hip).  Call after build_conv for the layers whose weights dwarf their activations.
inline void add_frag_packs(PlanBase& P, ConvL& L) {
    for (int i = 0; i < 4; ++i) {
        if (L.pk_fwd[i] >= 0) { PackDesc d = P.pk.d[L.pk_fwd[i]]; d.frag = 1; L.pk_fwd_f[i] = P.pk.add(d); }
        if (L.pk_dgrad[i] >= 0) { PackDesc d = P.pk.d[L.pk_dgrad[i]]; d.frag = 1; L.pk_dgrad_f[i] = P.pk.add(d); }
    }
}

// ------------------------------------------------------------------ launch helpers
// pkf: optional fragment-major copies of the same matrices (GatherClass::Wf)
inline GemmParams gemm_of(const PlanBase& P, const GatherPlan& pl, const int* pk, int groups, int group_n, const int* pkf = nullptr) {
    GemmParams g{};
    g.c = pl.c; g.c.groups = groups; g.c.group_n = group_n;
    g.npad = P.pk.d[pk[0]].Npad;
    int max_tiles = 0, min_nk = 1 << 30;
    for (int i = 0; i < pl.c.nclasses; ++i) {
        g.cls[i] = pl.cls[i];
        g.cls[i].rows_per_group = group_n * pl.cls[i].OY * pl.cls[i].OX;
        g.cls[i].Wp = P.buf.packed + P.pk.d[pk[i]].dst_off;
        g.cls[i].Wf = (pkf && pkf[i] >= 0) ? P.buf.packed + P.pk.d[pkf[i]].dst_off : nullptr;
        max_tiles = max(max_tiles, ceil_div(g.cls[i].rows_per_group, 128));
        min_nk = min(min_nk, ceil_div(g.cls[i].K, 64));
    }
    // few workgroups and a long K loop: split K so the chip is not idle behind a serial chain of tile latencies
    const int bn = pl.c.N <= 16 ? 16 : pl.c.N <= 32 ? 32 : pl.c.N <= 64 ? 64 : 128;
    const int tiles = max_tiles * groups * pl.c.nclasses * ceil_div(pl.c.N, bn);
    g.ksplit = 1;
    // (partial tiles go to per-split slabs with plain stores; a finish kernel sums them and runs the epilogue --
    //  float-atomic accumulation of the partial tiles measured slower than the latency chain it removed)
    if (!P.no_splitk && tiles <= 128 && min_nk >= 6) {
        int ks = min(min(8, min_nk / 3), max(1, 256 / tiles));
        if (ks > 1 && (size_t)tiles * ks * 128 * bn <= P.sk_floats) { g.ksplit = ks; g.sk_buf = P.sk_buf; g.sk_cnt = P.sk_cnt; }
    }
    return g;
}
inline WgradParams wgrad_of(const PlanBase& P, const GatherPlan& pl, const int* gk, int groups, int group_n) {
    WgradParams g{};
    g.c = pl.c; g.c.groups = groups; g.c.group_n = group_n;
    int max_rows = 0, max_kpad = 0;
    for (int i = 0; i < pl.c.nclasses; ++i) {
        g.cls[i] = pl.cls[i];
        g.cls[i].rows_per_group = group_n * pl.cls[i].OY * pl.cls[i].OX;
        g.cls[i].dWp = P.buf.gpk + P.gk.d[gk[i]].dst_off;
        g.cls[i].Kpad = P.gk.d[gk[i]].Kpad;
        max_rows = max(max_rows, groups * g.cls[i].rows_per_group);
        max_kpad = max(max_kpad, g.cls[i].Kpad);
    }
    const int tiles = ceil_div(pl.c.N, 64) * (max_kpad / 64) * pl.c.nclasses;
    int chunks = max(1, min(ceil_div(max_rows, 64), 768 / max(tiles, 1)));
    g.rows_per_block = round_up(ceil_div(max_rows, chunks), 64);
    return g;
}
inline GatherPlan dense_plan(int rows, int C, int ld, int N) { return plan_dense(rows, C, ld, N); }

// weight gradient of a (non-thin) ConvTranspose2d layer: `in_act` = the layer's activated input [groups*B][IH][IW][Cin],
// `d_out` = gradient w.r.t. its raw output [groups*B][OH][OW][Cout]
inline WgradParams convT_wgrad(const PlanBase& P, const ConvL& L, int groups, int B, const bf16* in_act, const bf16* d_out) {
    if (L.wgrad_fwdform) {
        WgradParams g = wgrad_of(P, L.dgrad, L.gk, groups, B);
        g.c.A = d_out; g.P = in_act; g.ldp = L.g.Cin;
        return g;
    }
    WgradParams g = wgrad_of(P, L.fwd, L.gk, groups, B);
    g.c.A = in_act; g.P = d_out; g.ldp = L.g.Cout;
    return g;
}

inline BnFinalizeArgs bn_fin_args(const PlanBase& P, const BnL& b, int rows_per_group, int G, const float2* stats, int updates,
                                  float2* aff, float2* mr, int training) {
    BnFinalizeArgs f{};
    f.stats = stats; f.G = G; f.C = b.C; f.count = (float)rows_per_group;
    f.gamma = P.buf.params + b.w_off; f.beta = P.buf.params + b.b_off;
    f.running_mean = P.buf.bn_stats + b.stat_off; f.running_var = P.buf.bn_stats + b.stat_off + b.C;
    f.num_batches_tracked = P.buf.bn_nbt + b.idx;
    f.updates_per_group = updates; f.affine = aff; f.meanrstd = mr; f.eps = BN_EPS; f.momentum = BN_MOM; f.training = training;
    f.skip_update_mask = G > 1 ? P.dec_skip_mask : 0u;
    return f;
}
inline int bn_act(PlanBase& P, const BnL& b, const bf16* r, bf16* a, int rows, int rows_per_group, int G, const float2* stats,
           int updates, float2* aff, float2* mr, int training, hipStream_t s) {
    BnActArgs x{};
    x.r = r; x.a = a; x.rows = rows; x.C = b.C; x.ld = b.C; x.rows_per_group = rows_per_group; x.G = G; x.act = ACT_SWISH;
    x.fin = bn_fin_args(P, b, rows_per_group, G, stats, updates, aff, mr, training);
    return launch_bn_act(x, s);
}
// The same normalise + activate pass for a layer whose CONSUMER staged the raw tensor itself (convres.hip, GatherTransform kind 1:
// tables written, running statistics updated there): only the weight gradient still wants the materialised operand, so this
// runs off the main chain, right in front of that weight gradient, and touches neither the tables nor the running buffers.
inline int bn_act_side(PlanBase& P, const BnL& b, const bf16* r, bf16* a, int rows, int rows_per_group, int G, const float2* stats,
                       int training, hipStream_t s) {
    BnActArgs x{};
    x.r = r; x.a = a; x.rows = rows; x.C = b.C; x.ld = b.C; x.rows_per_group = rows_per_group; x.G = G; x.act = ACT_SWISH;
    x.fin = bn_fin_args(P, b, rows_per_group, G, stats, 0, nullptr, nullptr, training);
    x.fin.running_mean = nullptr; x.fin.running_var = nullptr; x.fin.num_batches_tracked = nullptr;
    return launch_bn_act(x, s);
}


// ------------------------------------------------------------------ MLP layers (Linear [+ BatchNorm1d + activation])
struct MlpLin {
    long long w_off, b_off;
    int N, K;                   // out / in features
    int Kc;                     // operand row width (K padded to 8)
    int ldo;                    // output row stride (N padded to 8)
    int bn;                     // BN index or -1
    int pk_fwd, pk_dgrad, gk;
};
struct BnTabs { float2 *st, *red, *aff, *mr; };   // per-BatchNorm workspace tables: [G][SLOTS][C] x2, [G][C] x2

// Kc > 0 overrides the operand row width (e.g. the z operand [rows][ldz] whose column D holds a 1.0: it meets a
// zero weight column here)
inline void mlp_lin_init(PlanBase& P, MlpLin& L, const std::string& name, int N, int K, int bn, bool need_dgrad, int Kc = 0) {
    L.w_off = off(P, name + ".weight"); L.b_off = off(P, name + ".bias");
    L.N = N; L.K = K; L.Kc = Kc > 0 ? Kc : round_up(K, 8); L.ldo = round_up(N, 8); L.bn = bn;
    L.pk_fwd = P.pk.add(pack_dense(L.w_off, N, K, npad_for(N), round_up(L.Kc, 64), K, 1));
    L.gk = P.gk.add(pack_dense(L.w_off, N, K, round_up(N, 64), round_up(L.Kc, 64), K, 1));
    L.pk_dgrad = need_dgrad ? P.pk.add(pack_dense(L.w_off, K, N, npad_for(K), round_up(L.ldo, 64), 1, K)) : -1;
}
// y = A W^T + b  (A: [rows][L.Kc] bf16 activated operand).  groups > 1: BatchNorm groups of rows/groups rows.
inline int mlp_fwd(PlanBase& P, const MlpLin& L, const bf16* A, int rows, int groups, bf16* out_bf, float* out_f, float2* stats,
                   hipStream_t s) {
    GatherPlan pl = dense_plan(rows / groups, L.Kc, L.Kc, L.N);
    GemmParams g = gemm_of(P, pl, &L.pk_fwd, groups, rows / groups);
    g.c.A = A; g.bias = P.buf.params + L.b_off;
    g.out_bf = out_bf; g.out_f = out_f; g.ldo = out_f ? L.N : L.ldo; g.colstats = stats;
    return launch_gemm_gather(g, s);
}
// dA = dY W with the d-activation of the producer (`act` after the BatchNorm whose tables are `pbn`) fused; dY: [rows][L.ldo]
inline int mlp_dgrad(PlanBase& P, const MlpLin& L, const bf16* dY, int rows, int groups, bf16* out_bf, float* out_f, int out_ld,
                     const bf16* r_prev, const BnTabs* pbn, int act, hipStream_t s) {
    GatherPlan pl = dense_plan(rows / groups, L.ldo, L.ldo, L.K);
    GemmParams g = gemm_of(P, pl, &L.pk_dgrad, groups, rows / groups);
    g.c.A = dY; g.out_bf = out_bf; g.out_f = out_f; g.ldo = out_ld;
    if (r_prev) {
        g.d_r = r_prev; g.d_ld = out_ld; g.d_act = act;
        if (pbn) { g.d_affine = pbn->aff; g.d_meanrstd = pbn->mr; g.d_red = pbn->red; }
    }
    return launch_gemm_gather(g, s);
}
inline int mlp_wgrad(PlanBase& P, const MlpLin& L, const bf16* dY, const bf16* A, int rows, hipStream_t s) {
    GatherPlan pl = dense_plan(rows, L.Kc, L.Kc, L.N);
    WgradParams g = wgrad_of(P, pl, &L.gk, 1, rows);
    g.c.A = A; g.P = dY; g.ldp = L.ldo;
    return wgrad_async(P, g, s);
}
inline int bn1d_act(PlanBase& P, const BnL& b, const BnTabs& t, const bf16* r, bf16* a, int rows, int groups, int ld, int updates,
                    int training, int act, hipStream_t s) {
    BnActArgs x{};
    x.r = r; x.a = a; x.rows = rows; x.C = b.C; x.ld = ld; x.rows_per_group = rows / groups; x.G = groups; x.act = act;
    BnFinalizeArgs& f = x.fin;
    f.stats = t.st; f.G = groups; f.C = b.C; f.count = (float)(rows / groups);
    f.gamma = P.buf.params + b.w_off; f.beta = P.buf.params + b.b_off;
    f.running_mean = P.buf.bn_stats + b.stat_off; f.running_var = P.buf.bn_stats + b.stat_off + b.C;
    f.num_batches_tracked = P.buf.bn_nbt + b.idx;
    f.updates_per_group = updates; f.affine = t.aff; f.meanrstd = t.mr; f.eps = BN_EPS; f.momentum = BN_MOM; f.training = training;
    f.skip_update_mask = groups > 1 ? P.dec_skip_mask : 0u;
    return launch_bn_act(x, s);
}
inline int bn1d_bwd(PlanBase& P, const BnL& b, const BnTabs& t, bf16* d, const bf16* r, int rows, int groups, int ld, hipStream_t s) {
    BnBwdApplyArgs x{};
    x.db = d; x.r = r; x.dr = d; x.rows = rows; x.C = b.C; x.ld = ld; x.rows_per_group = rows / groups; x.G = groups;
    x.red = t.red; x.meanrstd = t.mr; x.gamma = P.buf.params + b.w_off;
    x.dgamma = P.buf.grads + b.w_off; x.dbeta = P.buf.grads + b.b_off;
    return launch_bn_bwd_apply(x, s);
}

inline hipEvent_t next_ev(PlanBase& P) {
    if (P.next_event == P.events.size()) {
        hipEvent_t e;
        // The step's events order streams of ONE device, and the kernels' own agent-scope release / acquire carry the data: the
        // system-scope fence an event performs by default when it completes (cache writeback + invalidation for the HOST's and
        // other devices' benefit) only slows the kernels behind it -- 671 -> 645 us per MultiMNIST step without it.  The host reads
        // results after a stream synchronisation, the gradient exchange runs behind kernels of the caller's stream.
        // (knob ev_flags: 0 the default fence, 2 none; hipEventReleaseToDevice alone changed nothing)
        const int ef = mmvae_knob("ev_flags", 2);
        hipEventCreateWithFlags(&e, hipEventDisableTiming | ((ef & 2) ? hipEventDisableSystemFence : 0u));
        P.events.push_back(e);
    }
    return P.events[P.next_event++];
}
// `to` waits for everything enqueued on `from` so far
inline int edge(PlanBase& P, hipStream_t from, hipStream_t to) {
    {   // measurement aid (only meaningful with the side work switched off): 1 all edges, 2 forks (main -> side), 3 joins
        const int k = mmvae_knob("dbg_skip_edges", 0);
        const bool to_side = to == P.st_text || to == P.st_wgrad || to == P.st_wgrad2;
        if (k == 1 || (k == 2 && to_side) || (k == 3 && !to_side)) return MMVAE_OK;
    }
    hipEvent_t e = next_ev(P);
    if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
        mmvae_set_error("stream fork/join failed: %s", hipGetErrorString(hipGetLastError()));
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}
inline int ensure_streams(PlanBase& P) {
    if (!P.st_text) {
        // side work (second-modality path, weight gradients) only feeds the optimizer at the end of the step and runs on
        // side streams.  The three side streams are shared by every plan of the process: ROCm gives a process 4 hardware queues, and
        // once more streams than that carry work they are time-sliced onto shared queues and every kernel slows down
        // several-fold (measured: 0.98 -> 2.78 ms per step with 7 streams).  Plans enqueue behind event edges, so sharing
        // streams between plans only serialises what would have been serialised anyway.
        static std::mutex mu;
        static hipStream_t shared_dev[32][3] = {};                  // per device: a second device gets streams of its own
        int cur_dev = 0;
        (void)hipGetDevice(&cur_dev);
        hipStream_t* shared = shared_dev[cur_dev & 31];
        std::lock_guard<std::mutex> g(mu);
        if (!shared[0]) {
            int least = 0, greatest = 0;
            hipDeviceGetStreamPriorityRange(&least, &greatest);
            // Default: every stream at the default priority.  Lowest-priority side streams (policy 0, opt-in through
            // mmvae_set_stream_policy) are worth <= 0.8 % (CelebA; nothing on MultiMNIST / COCO) and carry a cliff:
            // measured on MI355X / ROCm 7, as soon as ANOTHER default-priority stream carries work next to the main one
            // (an H2D copy stream, a collective library's internal stream) they make every kernel of the process run
            // several times slower (0.98 -> 2.63 ms per step).
            const bool low = mmvae_stream_policy() == 0;
            const bool flat = !low;
            mmvae_stream_policy_freeze();
            const int prio = flat ? 0 : least;
            for (int i = 0; i < 3; ++i)
                if (hipStreamCreateWithPriority(&shared[i], hipStreamNonBlocking, prio) != hipSuccess) {
                    mmvae_set_error("hipStreamCreate failed");
                    shared[0] = nullptr;
                    return MMVAE_EHIP;
                }
        }
        P.st_text = shared[0]; P.st_wgrad = shared[1]; P.st_wgrad2 = shared[2]; P.st_wgrad2_own = shared[2];
    }
    P.st_wgrad2 = (P.single_wgrad_stream || mmvae_knob("one_wgrad_stream", 0)) ? P.st_wgrad : P.st_wgrad2_own;
    P.next_event = 0; P.wgrad_rr = 0;
    return MMVAE_OK;
}

// End of a multi-stream step: the side streams meet on ONE of them first and the main stream waits once -- a wait on the main
// stream is a barrier packet in front of its next kernel (~5.5 us each, tools/step_parts.py), a wait between side streams costs
// the main chain nothing.
inline int join_sides(PlanBase& P, hipStream_t T, hipStream_t s) {
    hipStream_t hub = P.st_wgrad;
    // (with ONE weight-gradient stream there are two side streams: the main stream waiting for each of them directly measured
    //  628 against 634 us per MultiMNIST step)
    const int oj = mmvae_knob("one_join", -1);          // (-1: not set.  A call site caches the value it looked up: the default must not vary)
    if (!hub || hub == s || T == s || (oj >= 0 ? oj == 0 : P.st_wgrad2 == P.st_wgrad)) {
        if (T != s) MMVAE_TRY(edge(P, T, s));
        if (hub && hub != s) MMVAE_TRY(edge(P, hub, s));
        if (P.st_wgrad2 && P.st_wgrad2 != hub && P.st_wgrad2 != s) MMVAE_TRY(edge(P, P.st_wgrad2, s));
        if (P.spare_used && P.st_wgrad2_own && P.st_wgrad2_own != P.st_wgrad2 && P.st_wgrad2_own != s) MMVAE_TRY(edge(P, P.st_wgrad2_own, s));
        P.spare_used = false;
        return MMVAE_OK;
    }
    if (T != hub) MMVAE_TRY(edge(P, T, hub));
    if (P.st_wgrad2 && P.st_wgrad2 != hub) MMVAE_TRY(edge(P, P.st_wgrad2, hub));
    if (P.spare_used && P.st_wgrad2_own && P.st_wgrad2_own != P.st_wgrad2 && P.st_wgrad2_own != hub) MMVAE_TRY(edge(P, P.st_wgrad2_own, hub));
    P.spare_used = false;
    return edge(P, hub, s);
}

// Error exit of a multi-stream step: whatever was already forked onto the side streams is joined back into the caller's
// stream (so a caller that catches the error and frees or reuses the workspace does not race with work in flight) and
// the per-step scheduling state is reset.  Best effort: failures of the join itself are not reported over the first error.
inline void join_after_error(PlanBase& P, hipStream_t s) {
    if (P.st_text) {
        hipStream_t side[4] = {P.st_text, P.st_wgrad, P.st_wgrad2, P.st_wgrad2_own};
        for (int i = 0; i < 4; ++i) {
            if (!side[i] || side[i] == s || (i > 0 && side[i] == side[i - 1])) continue;
            hipEvent_t e = next_ev(P);
            if (hipEventRecord(e, side[i]) == hipSuccess) (void)hipStreamWaitEvent(s, e, 0);
        }
        (void)hipGetLastError();
    }
    P.deferred.clear();
    P.slab.jobs.clear(); P.slab.ring_jobs.clear();
    P.defer_wgrad = false; P.wgrad_forked = false; P.batch_reduce = false; P.spare_used = false; P.no_splitk = false; P.in_step = false;
    (void)mmvae_take_stop_event();
    P.side_pending.clear();
}

inline int check_bound(const PlanBase* P) {
    MMVAE_REQUIRE(P && P->bound, "plan has no buffers bound (mmvae_mm_bind)");
    return MMVAE_OK;
}


// out[16] = sum over the MMVAE_LOSS_SLOTS replicated rows of the loss accumulators.  alarm0 / alarm1 (optional): words a
// kernel of the step sets when it gave up waiting for a peer (the cluster exchange of the COCO caption decoder): the
// step's numbers are then garbage, and the loss sums say so (NaN) instead of looking plausible.
// A failed step must also never reach the parameters: `adam_state` (the caller's 16-byte optimizer state block, elementwise.h
// AdamArgs::step) gets its skip word set, and element 0 of the flat gradient becomes a NaN with the payload MMVAE_VOID_MARK -- a
// mark that survives the SUM all-reduce of a data-parallel job (NaN + x keeps the NaN's payload), so that EVERY rank's Adam
// kernel drops the update (adam_kernel).  A NaN of any other payload -- a genuinely diverged gradient -- is NOT a mark: it goes
// through the update and shows in the parameters, as with torch.optim.Adam in the reference (multimnist/train.py:173).
static __global__ void sum_slots_kernel(const float* slots, float* out, const unsigned* alarm0 = nullptr, const unsigned* alarm1 = nullptr,
                                        long long* adam_state = nullptr, float* grads = nullptr) {
    const int j = threadIdx.x;
    if (j >= 16) return;
    float s = 0.f;
    for (int q = 0; q < MMVAE_LOSS_SLOTS; ++q) s += slots[q * 16 + j];
    const bool alarm = (alarm0 && *alarm0) || (alarm1 && *alarm1);
    if (alarm) s = __builtin_nanf("");
    out[j] = s;
    if (alarm && j == 0) {
        if (adam_state) reinterpret_cast<unsigned*>(adam_state + 1)[1] = 1u;
        if (grads) grads[0] = __uint_as_float(MMVAE_VOID_MARK);
    }
}
static __global__ void cast_z_kernel(const float* z, int rows, int D, bf16* out, int ldz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ldz) return;
    int r = i / ldz, d = i - r * ldz;
    float v = d < D ? z[(size_t)r * D + d] : (d == D ? 1.0f : 0.0f);
    out[i] = (bf16)v;
}
static __global__ void cast_bf_kernel(const float* x, long long n, bf16* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16)x[i];
}
static __global__ void colsum_kernel(const float* x, int rows, int cols, float* out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(size_t)r * cols + c];
    out[c] += s;
}
// dlogit = d_recon * p * (1 - p)   (sigmoid backward for the drop-in decoder module)
static __global__ void sigmoid_bwd_kernel(const float* d_recon, const float* recon, long long n, float* dlogit) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float p = recon[i]; dlogit[i] = d_recon[i] * p * (1.0f - p); }
}
static __global__ void sigmoid_kernel(const float* logits, long long n, float* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.0f / (1.0f + expf(-logits[i]));
}

