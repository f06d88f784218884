"""The text decoder's forward kernel ALONE on the chip (module entry point, 768 rows = the fused step's 3 passes): phase stamps of
workgroup 0 and the kernel's duration -- the same kernel inside the step (tools/text_stamps.py) runs beside the image decoder."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd  # noqa
from multimodal_vae_amd import multimnist as M
from multimodal_vae_amd._lib import call
dev = torch.device('cuda:0')
torch.manual_seed(0)
vae = M.MultimodalVAE(n_latents=100, use_cuda=True).to(dev)
dec = vae.text_decoder
dec.train()
z = torch.randn(768, 100, device=dev)
for knob in (1, 0):
    call("mmvae_debug_set", b"text_fwd2", knob)
    for _ in range(3): dec(z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); e0.record(); dec(z); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"text_fwd2={knob}: module forward call (keep-mask kernel + decoder kernel + allocations) median {sorted(ts)[5]:.1f} us")
call("mmvae_debug_set", b"text_fwd2", 1)
tsb = torch.zeros(64, dtype=torch.int64, device=dev)
call("mmvae_debug_set", b"txt_ts_lo", ctypes.c_int(tsb.data_ptr() & 0xffffffff).value)
call("mmvae_debug_set", b"txt_ts_hi", ctypes.c_int(tsb.data_ptr() >> 32).value)
dec(z); torch.cuda.synchronize()
call("mmvae_debug_set", b"txt_ts_lo", 0); call("mmvae_debug_set", b"txt_ts_hi", 0)
t = tsb.cpu().double() / 100.0
n = int((t[:32] > 0).sum())
names = ["G0", "gates0", "G1", "gates1", "h2o", "softmax"]
print(f"alone: {n} stamps, loop {t[n - 1] - t[0]:.1f} us")
for i in range(4):
    row = [t[1 + i * 6 + k] - t[i * 6 + k] for k in range(6)]
    print(f"  step {i}: " + "  ".join(f"{names[k]} {row[k]:5.2f}" for k in range(6)) + f"   sum {sum(row):.2f}")
