"""Does a 5th active stream (what RCCL's internal stream is in the DP path) slow the step down?  modes: none | fake | rccl"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
mode = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
import multimodal_vae_amd
from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
from multimodal_vae_amd.init import default_init_

class Fake:
    def __init__(self): self.s = torch.cuda.Stream(priority=int(os.environ.get("FAKE_PRIO", "0")))
    def __call__(self, flat):
        cur = torch.cuda.current_stream()
        self.s.wait_stream(cur)
        with torch.cuda.stream(self.s): flat.mul_(1.0)
        cur.wait_stream(self.s)
class FakeOnce(Fake):
    def __init__(self): super().__init__(); self.n = 0
    def __call__(self, flat):
        self.n += 1
        if self.n <= 3: super().__call__(flat)
class Rccl:
    def __init__(self):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        self.dist = dist
    def __call__(self, flat): self.dist.all_reduce(flat)
ar = {"none": None, "fake": Fake, "once": FakeOnce, "rccl": Rccl}[mode]
ar = ar() if ar else None
B, D = 256, 100
state = MultimnistState(D, dev); default_init_(state, seed=1234)
image, text = bench.synthetic_batch(B, 1234)
image_d, text_d = image.to(dev), text.to(dev)
eng = FusedELBOStep(state, B, lr=1e-3, seed=1234, world_size=2 if ar else 1, all_reduce=ar)
for _ in range(30): eng(image_d, text_d)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): eng(image_d, text_d)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(mode, "GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES"), "ms/step %.4f" % (dt / 300 * 1e3), flush=True)
