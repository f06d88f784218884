"""Step rate of the other model families at their named batch sizes (SURVEY §8d configurations 1, 3, 5); the headline
MultiMNIST benchmark is ``bench.py`` at the repo root.

    python -m multimodal_vae_amd.bench_models celeba 512
    python -m multimodal_vae_amd.bench_models coco 128 --steps 10
    python -m multimodal_vae_amd.bench_models mnist 128 --precision bf16

Prints one JSON line: inputs resident in HBM, synthetic data, random-init weights, W warm-up steps then K timed steps
bracketed by ``torch.cuda.synchronize()``; a step = zero_grad + 3 passes + 3 losses + backward + Adam.
"""
from __future__ import annotations

import argparse
import json
import time

import torch


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("family", choices=("mnist", "celeba", "coco"))
    ap.add_argument("batch", type=int, nargs="?", default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fp32", help="mnist only: fp32 | bf16")
    args = ap.parse_args(argv)
    from .core import (CelebaState, CocoState, FusedCelebaStep, FusedCocoStep, FusedMnistStep, MnistState)
    from .init import default_init_
    dev = torch.device("cuda", torch.cuda.current_device())
    B = args.batch or {"mnist": 128, "celeba": 512, "coco": 128}[args.family]
    g = torch.Generator().manual_seed(0)
    if args.family == "celeba":
        st = CelebaState(100, dev); default_init_(st, 3)
        eng = FusedCelebaStep(st, B)
        a, b = torch.rand(B, 3, 64, 64, generator=g).to(dev), (torch.rand(B, 18, generator=g) < 0.3).float().to(dev)
        dtype, workload = "bf16", "celeba_64x64_conv_mmvae_18_attributes_3pass_elbo_step"
    elif args.family == "coco":
        st = CocoState(100, dev); default_init_(st, 3)
        eng = FusedCocoStep(st, B, 0.4 * torch.randn(300, generator=g))
        a, b = torch.rand(B, 3, 32, 32, generator=g).to(dev), (0.4 * torch.randn(B, 102, 300, generator=g)).to(dev)
        dtype, workload = "bf16 (image half) + f32 (caption GRUs)", "coco_32x32_conv_mmvae_glove_caption_gru_3pass_elbo_step"
    else:
        st = MnistState(20, dev, args.precision); default_init_(st, 3)
        eng = FusedMnistStep(st, B)
        a, b = torch.rand(B, 784, generator=g).to(dev), torch.randint(0, 10, (B,), generator=g).to(dev)
        dtype, workload = ("f32" if args.precision == "fp32" else "bf16"), "mnist_28x28_mlp_mmvae_label_3pass_elbo_step"
    for _ in range(args.warmup):
        out = eng(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res = {"metric": "ELBO-steps/sec, %s b=%d on 1 GPU" % (args.family, B), "value": 1.0 / dt, "unit": "ELBO-steps/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "samples_per_s": B / dt, "higher_is_better": True,
           "dtype": dtype, "data": "synthetic", "config": {"workload": workload, "batch_per_gpu": B},
           "workspace_gib": eng.ws.numel() / 2 ** 30, "final_losses": out.losses().cpu().tolist()}
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    main()
