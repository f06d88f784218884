"""Step time of the three model families at their named batch sizes (eager, multi-stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multimodal_vae_amd
from multimodal_vae_amd.core import CelebaState, FusedCelebaStep, MnistState, FusedMnistStep, CocoState, FusedCocoStep
from multimodal_vae_amd.init import default_init_
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "celeba"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"celeba": 512, "coco": 1024}.get(which, 128)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if which == "celeba":
    st = CelebaState(100, dev); default_init_(st, 3)
    eng = FusedCelebaStep(st, B)
    a = torch.rand(B, 3, 64, 64, device=dev); b = (torch.rand(B, 18, device=dev) < 0.3).float()
elif which == "coco":
    st = CocoState(100, dev); default_init_(st, 3)
    eng = FusedCocoStep(st, B, 0.4 * torch.randn(300))
    a = torch.rand(B, 3, 32, 32, device=dev); b = 0.4 * torch.randn(B, 102, 300, device=dev)
    print("workspace GiB", eng.ws.numel() / 2**30, flush=True)
else:
    st = MnistState(20, dev); default_init_(st, 3)
    eng = FusedMnistStep(st, B)
    a = torch.rand(B, 784, device=dev); b = torch.randint(0, 10, (B,), device=dev)
for _ in range(5):
    out = eng(a, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = eng(a, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{which} B={B}: {dt*1e3:.3f} ms/step  {1/dt:.1f} steps/s  {B/dt:.0f} samples/s  losses {out.losses().cpu().numpy()}")
if os.environ.get("TRY_GRAPH"):
    eng.capture(a, b)
    for _ in range(5):
        eng.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = eng.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{which} B={B} GRAPH: {dt*1e3:.3f} ms/step  {1/dt:.1f} steps/s  losses {out.losses().cpu().numpy()}")
