"""Does the DeviceBatcher's copy stream slow the step down (stream-priority effect)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
from multimodal_vae_amd import data as D
from multimodal_vae_amd._lib import call
mode = sys.argv[1]
dev = torch.device("cuda", 0)
B = 256
if mode == "flat":
    call("mmvae_set_stream_policy", 1)
state = MultimnistState(100, dev); default_init_(state, seed=1234)
eng = FusedELBOStep(state, B, lr=1e-3, seed=1234)
x, y = D.synthetic_multimnist(B * 40, seed=1)
from multimodal_vae_amd.utils import charlist_tensor
t = torch.stack([charlist_tensor(l) for l in y])
loader = D.DeviceBatcher(x, t, B, dev, shuffle=True, seed=0)
for epoch in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for im, tx in loader:
        eng(im, tx); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "epoch", epoch, "ms/step %.4f" % (dt / n * 1e3), flush=True)
