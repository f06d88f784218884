import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_
dev = torch.device("cuda:0")
for B in (1024, 2048, 4096, 8192):
    try:
        st = MultimnistState(100, dev); default_init_(st, 1)
        eng = FusedELBOStep(st, B)
        img, txt = bench.synthetic_batch(B, 5)
        img, txt = img.to(dev), txt.to(dev)
        for _ in range(3): out = eng(img, txt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = eng(img, txt)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(B, "ms/step %.3f" % (dt * 1e3), "samples/s %.0f" % (B / dt), out.losses().cpu().numpy(), "ws GiB %.2f" % (eng.ws.numel() / 2**30), flush=True)
    except Exception as e:
        print(B, "ERROR", str(e)[:200], flush=True)
