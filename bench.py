"""ELBO-steps/sec of the MultiMNIST 3-pass MMVAE training step (multimnist/train.py:146-173) on MI355X.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = zero_grad -> 3 passes -> 3 losses -> backward -> (grad all-reduce) -> Adam on a synthetic batch of 256
samples per GPU that is resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # read by ROCr at hsa_init: before ANY GPU call (dp.ensure_ipc_env)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 427.24e6          # reference fwd+bwd FLOPs per sample per ELBO step (FlopCounterMode)
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def synthetic_batch(B, seed):
    """U[0,1) image masked to ~15 % non-zero, digits with FILL(11)-padded suffix."""
    rng = np.random.default_rng(seed)
    img = rng.random((B, 1, 50, 50), dtype=np.float32)
    img *= (rng.random((B, 1, 50, 50)) < 0.15)
    text = rng.integers(0, 10, size=(B, 4)).astype(np.int64)
    lens = rng.integers(0, 5, size=(B,))
    for b in range(B):
        text[b, lens[b]:] = 11
    return torch.from_numpy(img), torch.from_numpy(text)


# HBM-side bytes per launch of the roofline kernel from rocprofv3 PMC passes (one pass per counter, no trace domains):
# FETCH_SIZE 30,725 KB x 2 (gfx950 tallies the 128-B requests of 16-B/lane reads at 64 B: MI355X_MICROARCH "HBM") +
# WRITE_SIZE 32,802 KB.  Raw rows: profiles/r01_pmc_{fetch,write}_size_dec_convT3.csv.  Algorithmic bytes of the
# layer: 14.2 MB activations in + 30.7 MB raw output = 44.9 MB; the extra reads are the 8 per-XCD L2s each fetching
# their own copy of input rows (the kernel is MFMA/issue-bound, not HBM-bound: 94 MB / 57 us = 1.6 TB/s).
MEASURED_TRAFFIC = {("dec_convT3", 256, 100): 2 * 30725e3 + 32802e3}


def cpu_baseline(B, D, image, text, budget_s=20.0):
    """The oracle (CPU restatement pinned to the reference) timed on this host's cores: reported, not the target."""
    from oracle import mmvae_ref as R
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    if hasattr(os, "sched_getaffinity"):
        cores = min(cores, len(os.sched_getaffinity(0)))
    try:                                                   # container CPU share (cgroup v2 / v1)
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            cores = max(1, min(cores, int(int(q[0]) / int(q[1]))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cores = max(1, min(cores, quota // period))
        except Exception:
            pass
    cores = int(os.environ.get("MMVAE_CPU_BASELINE_THREADS", min(cores, 64)))
    torch.set_num_threads(cores)
    P = R.formula_params("multimnist", D, requires_grad=True)
    names = [n for n, _ in R.param_table("multimnist", D)]
    plist = [P[n] for n in names]
    m = [torch.zeros_like(p) for p in plist]
    v = [torch.zeros_like(p) for p in plist]
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        for p in plist:
            p.grad = None
        losses, _ = R.multimnist_step_losses(P, image, text, True, 1e-3)
        (losses[0] + losses[1] + losses[2]).backward()
        step += 1
        R.adam_step(plist, [p.grad for p in plist], m, v, step)
        dt = time.perf_counter() - t0
        if step > 1:
            times.append(dt)
        if (len(times) >= 3 and time.perf_counter() - t_start > budget_s) or len(times) >= 12:
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "ELBO-steps/s", "cores": int(cores), "kind": "port",
            "sample": "%d full 3-pass fwd+bwd+Adam steps at B=%d after 1 warm-up (median), oracle/mmvae_ref.py, fp32, torch %s CPU"
                      % (len(times), B, torch.__version__),
            "samples_per_s": B / med}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--n_latents", type=int, default=100)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (1 GPU only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-layer", default="dec_convT3")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd.core import MultimnistState, FusedELBOStep
    from multimodal_vae_amd.init import default_init_
    from multimodal_vae_amd._lib import call

    from multimodal_vae_amd import dp
    all_reduce = None
    if world > 1:
        dp.init_distributed("nccl", dev)                   # backend "nccl" IS RCCL on ROCm
        all_reduce = dp.GradAllReduce()                     # one SUM all-reduce of the 9.35 MB flat gradient per step

    B, D = args.batch, args.n_latents
    state = MultimnistState(D, dev)
    default_init_(state, seed=1234)                       # identical replicas on every rank
    dp.broadcast_flat(state.params)
    image, text = synthetic_batch(B, dp.rank_seed(1234, rank))
    image_d, text_d = image.to(dev), text.to(dev)
    eng = FusedELBOStep(state, B, lr=1e-3, seed=dp.rank_seed(1234, rank), world_size=world, all_reduce=all_reduce)

    # the step is already ONE host call that enqueues ~75 kernels on three streams; graph replay is optional
    use_graph = args.graph and world == 1
    if use_graph:
        eng.capture(image_d, text_d)
        run = eng.replay
    else:
        run = lambda: eng(image_d, text_d)               # noqa: E731

    def barrier():
        dp.barrier(dev)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    import gc
    gc.collect()
    gc.disable()            # a generational collection in the enqueue thread stalls the GPU for milliseconds
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = out.losses().cpu().numpy().tolist()
    steps_per_s = args.steps / dt

    result = {
        "metric": "ELBO-steps/sec (whole node), MultiMNIST b=256 per GPU",
        "value": steps_per_s, "unit": "ELBO-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "samples_per_s": steps_per_s * B * world,
        "config": {"workload": "multimnist_50x50_conv_mmvae_gru_text_3pass_elbo_step", "batch_per_gpu": B,
                   "global_batch": B * world, "n_latents": D, "parallelism": "dp%d" % world,
                   "hip_graph": use_graph, "optimizer": "adam_lr1e-3",
                   "mfma": "bf16 in / fp32 acc; fp32 master weights, BN stats, PoE/KL/BCE/NLL, Adam"},
        "final_losses": losses,
        "step_mfma_frac": steps_per_s / world * B * FLOP_PER_SAMPLE / (PEAK_BF16_TFLOPS * 1e12),
    }

    # ---- per-step distribution: HIP events around single steps, outside the timed region
    n_ev = min(100, args.steps)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
    evs[0].record()
    for i in range(n_ev):
        run()
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_ev))
    result["ms_per_step_p10_p50_p90"] = [per[int(0.1 * (n_ev - 1))], per[(n_ev - 1) // 2], per[int(0.9 * (n_ev - 1))]]

    if rank == 0:
        # ---- roofline of the dominant GEMM kernel, timed live with HIP events on its own stream
        layer = args.roofline_layer.encode()
        flops = call("mmvae_mm_layer_flops", eng.h, layer)
        iters = 50
        s = torch.cuda.current_stream()
        st = __import__("ctypes").c_void_p(s.cuda_stream)
        call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer, 5, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer, iters, st)
        e1.record(s)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        result["roofline"] = {"kernel": "gemm_gather_kernel (%s)" % args.roofline_layer, "bound": "mfma",
                              "achieved": flops / (us * 1e-6) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "frac": flops / (us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, "us_per_launch": us,
                              "flops_per_launch": flops,
                              "traffic": MEASURED_TRAFFIC.get((args.roofline_layer, B, D))}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(B, D, image, text)
            result["speedup_vs_cpu_baseline"] = steps_per_s / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
