"""Phase stamps of the MultiMNIST text decoder's forward kernel (csrc/text.hip text_decoder_fwd2_kernel, workgroup 0): s_memrealtime
(100 MHz) after every barrier-delimited phase of every time step, inside a real step.  usage: python tools/text_stamps.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd  # noqa
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
from bench import synthetic_batch
dev = torch.device('cuda:0'); B = 256
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234)
img, txt = img.to(dev), txt.to(dev)
eng = FusedELBOStep(st, B)
for _ in range(5): eng(img, txt)
torch.cuda.synchronize()
names = ["G0 (gi0, gh0 GEMMs)", "gates0 + dropout", "G1 (gi1, gh1 GEMMs)", "gates1 + copy", "h2o GEMM", "softmax + embed"]
for alone in (0, 1):
    tsb = torch.zeros(64, dtype=torch.int64, device=dev)
    call("mmvae_debug_set", b"txt_ts_lo", ctypes.c_int(tsb.data_ptr() & 0xffffffff).value)
    call("mmvae_debug_set", b"txt_ts_hi", ctypes.c_int(tsb.data_ptr() >> 32).value)
    if alone: call("mmvae_debug_set", b"dbg_skip_wgrad", 1)
    eng(img, txt); torch.cuda.synchronize()
    call("mmvae_debug_set", b"txt_ts_lo", 0); call("mmvae_debug_set", b"txt_ts_hi", 0); call("mmvae_debug_set", b"dbg_skip_wgrad", 0)
    t = tsb.cpu().double() / 100.0
    n = int((t[:32] > 0).sum())
    print(f"--- {'without weight gradients' if alone else 'full step'}: {n} stamps, loop {t[n - 1] - t[0]:.1f} us")
    for i in range(4):
        row = [t[1 + i * 6 + k] - t[i * 6 + k] for k in range(6)]
        print(f"  step {i}: " + "  ".join(f"{names[k].split(' ')[0]} {row[k]:5.2f}" for k in range(6)) + f"   sum {sum(row):.2f}")
