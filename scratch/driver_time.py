import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodal_vae_amd
from multimodal_vae_amd import train as T
t0 = time.perf_counter()
T.main(["--cuda", "--epochs", "3", "--synthetic", "32768", "--batch_size", "256", "--log_interval", "50", "--out", "/tmp/ck"])
print("driver total s", time.perf_counter() - t0)
