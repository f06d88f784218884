"""GPU check of the ring-staged weight-gradient kernel (csrc/wgrad_ring.hip) against the streamed kernel (gemm.hip: wgrad_kernel)
on the activations / gradients of a real B=256 step: the same layer launched through mmvae_mm_bench_layer with the ring path off
and on, the packed fp32 gradients compared element by element, both timed (kernel + its reduce launch, as in the step).
usage: python tools/wgrad_check.py [B] [layer ...] [knob=value ...]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
sys.path.insert(0, '.')
from bench import synthetic_batch

dev = torch.device('cuda:0')
args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 256
knobs = [a for a in args if "=" in a]
sel = [a for a in args if "=" not in a] or ['dec_convT3_wgrad', 'dec_convT2_wgrad', 'enc_conv2_wgrad', 'enc_conv3_wgrad', 'dec_convT1_wgrad', 'enc_conv4_wgrad']
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234)
eng = FusedELBOStep(st, B)
eng(img.to(dev), txt.to(dev)); st.ensure_packed(); torch.cuda.synchronize()
for kv in knobs:
    k, v = kv.split("="); call("mmvae_debug_set", k.encode(), int(v))
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)

def run(layer, ring, iters):
    call("mmvae_debug_set", b"wgrad_ring", ring)
    call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer.encode(), iters, sp)

bad = 0
for L in sel:
    res = []
    for ring in (0, 1):
        st.gpk.zero_()
        run(L, ring, 1); torch.cuda.synchronize()
        g = st.gpk.clone()
        run(L, ring, 3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(L, ring, 20); e1.record(s); torch.cuda.synchronize()
        res.append((g, e0.elapsed_time(e1) * 1e3 / 20))
    (g0, t0), (g1, t1) = res
    fl = call("mmvae_mm_layer_flops", eng.h, L.encode())
    nz = int((g0 != 0).sum())
    den = g0.abs().max().item() + 1e-30
    err = (g0 - g1).abs().max().item() / den
    rel = ((g0 - g1).norm() / (g0.norm() + 1e-30)).item()
    ok = rel < 2e-5 and nz > 0 and bool(torch.isfinite(g1).all())
    bad += 0 if ok else 1
    print(f"{L:18s} streamed {t0:7.1f} us  ring {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TFLOP/s)  nonzero {nz}  max-err {err:.2e} rel-l2 {rel:.2e}  "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        d = (g0 - g1).abs()
        idx = torch.nonzero(d > 1e-3 * den).flatten()
        print(f"   {idx.numel()} elements differ; first: {idx[:8].tolist()}  g0 {g0[idx[:4]].tolist()}  g1 {g1[idx[:4]].tolist()}")
    if os.environ.get("WR_DBG"):
        ts = []
        for dbg in (1, 2, 4, 6, 7):
            call("mmvae_debug_set", b"wr_dbg", dbg)
            run(L, 1, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); run(L, 1, 20); e1.record(s); torch.cuda.synchronize()
            ts.append(f"dbg{dbg}={e0.elapsed_time(e1) * 1e3 / 20:.1f}")
        call("mmvae_debug_set", b"wr_dbg", 0)
        print("      (1 no stores, 2 no MFMA loop, 4 no DMA traffic)  " + "  ".join(ts), flush=True)
    if os.environ.get("WR_TS"):
        tsb = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        call("mmvae_debug_set", b"wr_ts_lo", ctypes.c_int(tsb.data_ptr() & 0xffffffff).value)
        call("mmvae_debug_set", b"wr_ts_hi", ctypes.c_int(tsb.data_ptr() >> 32).value)
        run(L, 1, 1); torch.cuda.synchronize()
        call("mmvae_debug_set", b"wr_ts_lo", 0); call("mmvae_debug_set", b"wr_ts_hi", 0)
        t = tsb.view(-1, 8).cpu()
        t = t[t[:, 0] > 0].double()
        base = t[:, 0].min()
        mhz = float(os.environ.get("WR_MHZ", "100"))          # s_memtime ticks per microsecond (100 MHz constant clock on gfx950)
        print(f"      {t.shape[0]} workgroups; span {(t[:, 5].max() - base) / mhz:.1f} us; last start +{(t[:, 0].max() - base) / mhz:.1f} us; "
              f"lifetime mean {((t[:, 5] - t[:, 0]) / mhz).mean():.1f} max {((t[:, 5] - t[:, 0]) / mhz).max():.1f} us")
        for cls in range(4):
            m = t[:, 7] == cls
            if not m.any(): continue
            x = t[m]
            iss = (x[:, 6].long() >> 16).double(); x[:, 6] = (x[:, 6].long() & 0xffff).double()
            print(f"      class {cls}: {int(m.sum())} wgs, batches {x[:, 6].mean():.1f}; DMA issue {(iss / mhz).mean():.2f}; setup {((x[:, 1] - x[:, 0]) / mhz).mean():.2f}  first fill wait "
                  f"{((x[:, 2] - x[:, 1]) / mhz).mean():.2f}  loop {((x[:, 4] - x[:, 2]) / mhz).mean():.2f} (of it waiting {(x[:, 3] / mhz).mean():.2f})  "
                  f"epilogue {((x[:, 5] - x[:, 4]) / mhz).mean():.2f} us", flush=True)
    if os.environ.get("WR_WGS"):
        ts = []
        for wgs in (1, 2, 3, 4, 6, 8):
            call("mmvae_debug_set", b"wr_wgs", wgs)
            run(L, 1, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); run(L, 1, 20); e1.record(s); torch.cuda.synchronize()
            ts.append(f"wgs{wgs}={e0.elapsed_time(e1) * 1e3 / 20:.1f}")
        call("mmvae_debug_set", b"wr_wgs", 4)
        print("      workgroups per 4 CUs:  " + "  ".join(ts), flush=True)
print("FAILED" if bad else "all layers agree")
sys.exit(1 if bad else 0)
