"""GPU check of the image-resident conv kernels (csrc/convres.hip) against the generic gather GEMM on the activations of a
real B=256 step: same layer launched through mmvae_mm_bench_layer with the path off and on, outputs and column sums
compared, both timed.  usage: python tools/convres_check.py [B]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
sys.path.insert(0, '.')
from bench import synthetic_batch

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234)
eng = FusedELBOStep(st, B)
call("mmvae_debug_set", b"convres", 0)
eng(img.to(dev), txt.to(dev)); st.ensure_packed(); torch.cuda.synchronize()
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
SLOTS = 16

def view(name, nbytes, dtype):
    off = call("mmvae_mm_debug_offset", eng.h, name.encode())
    assert off >= 0, name
    return eng.ws[off:off + nbytes].view(dtype)

# layer -> (output buffer, images, pixels, channels, stats buffer, stat groups)
G3 = 3
layers = {
    'enc_conv2': ('r2', B, 144, 64, 'st_e0', 1),
    'enc_conv3': ('r3', B, 36, 128, 'st_e1', 1),
    'dec_convT2': ('q2', 3 * B, 144, 64, 'st_d1', 3),
    'dec_convT3': ('q3', 3 * B, 625, 32, 'st_d2', 3),
    'enc_conv2_dgrad': ('d1e', B, 625, 32, None, 1),
    'enc_conv3_dgrad': ('d2e', B, 144, 64, 'red_e0', 1),
    'dec_convT2_dgrad': ('d1', 2 * B, 36, 128, 'red_d0', 2),
    'dec_convT3_dgrad': ('d2', 2 * B, 144, 64, 'red_d1', 2),
    'enc_conv4': ('r4', B, 4, 256, 'st_e2', 1),
    'dec_convT1': ('q1', 3 * B, 36, 128, 'st_d0', 3),
    'enc_conv4_dgrad': ('d3e', B, 36, 128, 'red_e1', 1),
    'dec_convT1_dgrad': ('du', 2 * B, 4, 256, None, 2),
}
sel = sys.argv[2:] or list(layers)

def run(layer, on, iters):
    call("mmvae_debug_set", b"convres", on)
    call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer.encode(), iters, sp)

bad = 0
for L in sel:
    out_name, nimg, pix, ch, st_name, groups = layers[L]
    out = view(out_name, nimg * pix * ch * 2, torch.bfloat16)
    stats = view(st_name, G3 * SLOTS * ch * 8, torch.float32) if st_name else None
    res = []
    for on in (0, 1):
        out.zero_()
        if stats is not None: stats.zero_()
        run(L, on, 1); torch.cuda.synchronize()
        o = out.float().clone()
        sres = stats.view(G3, SLOTS, ch, 2).sum(1).clone() if stats is not None else None
        run(L, on, 3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(L, on, 20); e1.record(s); torch.cuda.synchronize()
        res.append((o, sres, e0.elapsed_time(e1) * 1e3 / 20))
    (o0, s0, t0), (o1, s1, t1) = res
    fl = call("mmvae_mm_layer_flops", eng.h, L.encode())
    den = o0.abs().max().item() + 1e-30
    err = (o0 - o1).abs().max().item() / den
    rel = ((o0 - o1).norm() / (o0.norm() + 1e-30)).item()
    serr = 0.0
    if s0 is not None:
        serr = ((s0 - s1).abs().max() / (s0.abs().max() + 1e-30)).item()
    ok = rel < 4e-3 and serr < 2e-3 and o1.abs().max().item() > 0
    bad += 0 if ok else 1
    print(f"{L:18s} old {t0:7.1f} us  new {t1:7.1f} us  ({fl / t1 / 1e6:6.1f} TFLOP/s)  out max-err {err:.2e} rel-l2 {rel:.2e}  "
          f"stats err {serr:.2e}  {'ok' if ok else 'MISMATCH'}", flush=True)
    if os.environ.get("CR_ALT"):
        ts = []
        for alt in (1, 5):
            call("mmvae_debug_set", b"convres_alt", alt)
            out.zero_()
            run(L, 1, 3); torch.cuda.synchronize()
            rel2 = ((o0 - out.float()).norm() / (o0.norm() + 1e-30)).item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); run(L, 1, 20); e1.record(s); torch.cuda.synchronize()
            ts.append(f"alt{alt}={e0.elapsed_time(e1) * 1e3 / 20:.1f} (rel {rel2:.1e})")
        call("mmvae_debug_set", b"convres_alt", 0)
        print("      " + "  ".join(ts), flush=True)
    if os.environ.get("CR_TS"):
        tsb = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
        call("mmvae_debug_set", b"convres_ts_lo", ctypes.c_int(tsb.data_ptr() & 0xffffffff).value)
        call("mmvae_debug_set", b"convres_ts_hi", ctypes.c_int(tsb.data_ptr() >> 32).value)
        run(L, 1, 1); torch.cuda.synchronize()
        call("mmvae_debug_set", b"convres_ts_lo", 0); call("mmvae_debug_set", b"convres_ts_hi", 0)
        t = tsb.view(-1, 16).cpu()
        t = t[t[:, 0] > 0].double()
        names = ["start", "zero+tab+issue", "barrier", "stage+w0", "barrier", "c0 mfma", "c0 epi", "c1 mfma", "c1 epi", "c2 mfma", "c2 epi",
                 "c3 mfma", "c3 epi", "to end of classes", "atomics"]
        base = t[:, 0].min()
        print(f"      stamps of {t.shape[0]} waves; kernel span {(t[:, 14].max() - base) / 100:.1f} us; last wave start +{(t[:, 0].max() - base) / 100:.1f} us; "
              f"wave lifetime mean {((t[:, 14] - t[:, 0]) / 100).mean():.1f} max {((t[:, 14] - t[:, 0]) / 100).max():.1f} us")
        prev = t[:, 0]
        line = []
        for i in range(1, 15):
            cur = t[:, i]
            okm = cur > 0
            if okm.any():
                d = (cur[okm] - prev[okm]) / 100
                line.append(f"{names[i]} {d.mean():.2f}")
                prev = torch.where(okm, cur, prev)
        print("      mean us per phase: " + " | ".join(line), flush=True)
    if os.environ.get("CR_DBG"):
        ts = []
        for dbg in (1, 2, 4, 3, 7):
            call("mmvae_debug_set", b"convres_dbg", dbg)
            run(L, 1, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); run(L, 1, 20); e1.record(s); torch.cuda.synchronize()
            ts.append(f"dbg{dbg}={e0.elapsed_time(e1) * 1e3 / 20:.1f}")
        call("mmvae_debug_set", b"convres_dbg", 0)
        print("      (1 no stores, 2 no MFMA loop, 4 no staging)  " + "  ".join(ts), flush=True)
    if not ok:
        d = (o0 - o1).abs().view(nimg, pix, ch)
        print("   worst image/pixel/channel:", [int(x) for x in torch.nonzero(d == d.max())[0]],
              " per-image max err of first 6:", [round(d[i].max().item() / den, 4) for i in range(min(6, nimg))])
print("FAILED" if bad else "all layers agree")
sys.exit(1 if bad else 0)
