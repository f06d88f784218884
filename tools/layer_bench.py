import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd._lib import call
sys.path.insert(0, '.')
from bench import synthetic_batch
dev = torch.device('cuda:0'); B = 256
st = MultimnistState(100, dev); default_init_(st, 1234)
img, txt = synthetic_batch(B, 1234)
eng = FusedELBOStep(st, B)
eng(img.to(dev), txt.to(dev)); st.ensure_packed(); torch.cuda.synchronize()
for kv in [a for a in sys.argv[1:] if "=" in a]:          # knob=value arguments (mmvae_debug_set)
    k, v = kv.split("="); call("mmvae_debug_set", k.encode(), int(v))
layers = [a for a in sys.argv[1:] if "=" not in a] or ['enc_conv1', 'enc_conv2', 'enc_conv3', 'enc_conv4', 'enc_conv2_dgrad', 'enc_conv3_dgrad', 'enc_conv4_dgrad', 'dec_convT1', 'dec_convT2', 'dec_convT3', 'dec_convT1_dgrad', 'dec_convT2_dgrad', 'dec_convT3_dgrad', 'enc_fc1', 'enc_fc2', 'enc_fc3', 'enc_fc3_dgrad', 'enc_fc2_dgrad', 'dec_up', 'dec_up_dgrad', 'dec_convT1_wgrad', 'dec_convT2_wgrad', 'dec_convT3_wgrad', 'dec_last_wgrad', 'enc_conv1_wgrad', 'enc_conv2_wgrad', 'enc_conv3_wgrad', 'enc_conv4_wgrad']
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
for L in layers:
    fl = call("mmvae_mm_layer_flops", eng.h, L.encode())
    call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), L.encode(), 3, sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), L.encode(), 20, sp); e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{L:20s} {us:8.1f} us  {fl/1e9:7.2f} GFLOP  {fl/us/1e6:7.1f} TFLOP/s")
