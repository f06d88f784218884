"""A/B timing of the fused MultiMNIST step under sets of library knobs (mmvae_debug_set), interleaved rounds in ONE process
(cdna_hip_programming.md rule 24).  usage: python tools/step_ab.py name:knob=v,knob=v name2:... [rounds=5] [n=200]
A knob that a configuration does not name is reset to the value given for it in the FIRST configuration (or 0)."""
import os, sys, time, torch
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
import gc; gc.disable()
rounds, n = 5, 200
cfgs = []
for a in sys.argv[1:]:
    if a.startswith("rounds="): rounds = int(a[7:]); continue
    if a.startswith("n="): n = int(a[2:]); continue
    name, _, kv = a.partition(":")
    cfgs.append((name, dict((k, int(v)) for k, v in (x.split("=") for x in kv.split(",") if x))))
keys = sorted({k for _, d in cfgs for k in d})
base = {k: cfgs[0][1].get(k, 0) for k in keys}

def timeit():
    for _ in range(20): eng(img, txt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): eng(img, txt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

res = {name: [] for name, _ in cfgs}
for r in range(rounds):
    for name, d in cfgs:
        for k in keys: call("mmvae_debug_set", k.encode(), d.get(k, base[k]))
        res[name].append(timeit())
for name, _ in cfgs:
    v = sorted(res[name])
    print(f"{name:28s} median {v[len(v)//2]:8.1f}  min {v[0]:8.1f}  max {v[-1]:8.1f} us/step", flush=True)
