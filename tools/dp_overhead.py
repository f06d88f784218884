"""Single-GPU cost of the data-parallel branch: a world-size-1 RCCL group with the collective forced on.
Prints ms/step of the packed single-GPU path, the plain DP path (full unpack, one all-reduce, Adam) and the overlapped
DP path (early/late unpack, range all-reduces on a communication stream).  usage: python tools/dp_overhead.py [steps]"""
import os, sys, socket, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("DP_OVERHEAD_HWQ", "8"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import multimodal_vae_amd  # noqa: F401
from multimodal_vae_amd import dp
from multimodal_vae_amd.core import FusedELBOStep, MultimnistState
from multimodal_vae_amd.init import default_init_
from bench import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
modes = [a for a in sys.argv[2:] if "=" not in a] or ["packed", "dp", "dp_overlap"]
from multimodal_vae_amd._lib import call
for kv in [a for a in sys.argv[2:] if "=" in a]:              # knob=value (mmvae_debug_set)
    k, v = kv.split("="); call("mmvae_debug_set", k.encode(), int(v))
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
no_pg = os.environ.get("DP_OVERHEAD_NO_PG") is not None       # reference: the same loop without any process group
if not no_pg:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
img, txt = synthetic_batch(256, 1234)
img, txt = img.to(dev), txt.to(dev)
import gc
gc.disable()
for mode in (("packed",) if no_pg else modes):
    st = MultimnistState(100, dev); default_init_(st, 1234)
    ar = None if mode == "packed" else dp.GradAllReduce(force=True, overlap=mode == "dp_overlap")
    eng = FusedELBOStep(st, 256, world_size=1, all_reduce=ar)
    for _ in range(30):
        eng(img, txt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng(img, txt)
    torch.cuda.synchronize()
    print("%-11s %.4f ms/step" % (mode, (time.perf_counter() - t0) * 1e3 / steps))
if not no_pg:
    dist.destroy_process_group()
