"""ELBO-steps/sec of the 3-pass MMVAE training step on MI355X (default: MultiMNIST b=256, multimnist/train.py:146-173).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --workload celeba        (config 3: CelebA 64x64 b=512, celeba/train.py:131-147)
  python bench.py --workload coco          (config 5's per-GPU share: COCO 32x32 + 102-step captions b=128, coco/train.py:138-173)

One "step" = zero_grad -> 3 passes -> 3 losses -> backward -> (grad all-reduce) -> Adam on a synthetic batch that is
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

The JSON line carries, next to the contract fields:
  step_mfma_frac   reference FLOP count (FlopCounterMode on the reference) x steps/s / dense bf16 MFMA peak
  executed_flops_per_step  GEMM FLOPs this engine actually enqueues per step (encoder passes deduplicated, zero-padded taps
                   included; counted by the launchers), and step_mfma_frac_executed on that count
  step_hbm_frac    algorithmic bytes per step (0.59 GB at B=256) x steps/s / 8 TB/s
  roofline         the GEMM kernel FAMILY with the largest GPU-time share of the step, measured IN THE STEP: every launch of 20
                   extra steps carries its own HIP start/stop events (mmvae_debug_probe); achieved = sum of the family's
                   ALGORITHMIC FLOPs / sum of the durations of ALL its launches, with the other streams' kernels running beside
                   them.  A family is what does one job together: the weight-gradient kernels (wgrad_ring_kernel, wgrad_kernel,
                   wgrad_multi_kernel) AND the reduce launches that sum their partial copies are one family -- the reduce launches
                   carry time and no FLOPs.  `launches`: the per-launch numbers.  `frac_isolated*`: the same launches replayed
                   alone (mmvae_mm_bench_layer).  `traffic`: PMC-measured HBM-side bytes of the family's longest launch
                   (profiles/r04_traffic.json) next to its algorithmic bytes.  Agrees with the rocprofv3 --kernel-trace --stats
                   summary of this command under profiles/r04_<workload>_kernel_stats.csv
  roofline_second  the same for the runner-up family (the image-resident conv kernel `convres_kernel` or the weight gradients)
  kernels_in_step  every probed kernel of the step (all workloads): launches per step, in-step microseconds, TFLOP/s
  roofline_wgrad / roofline_dgrad   weight- and data-gradient launches of the last 64->32 layer replayed alone (as in round 2)
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # read by ROCr at hsa_init: before ANY GPU call (dp.ensure_ipc_env)
# The step uses up to 4 streams and an RCCL process group brings its own: past the runtime's default of 4 hardware queues
# the streams are time-sliced onto shared queues and EVERY kernel slows down (measured with a world-1 RCCL group:
# 0.90 -> 1.20 ms per step; with 8 queues 0.90 -> 0.91).  Read when the HIP runtime initialises: before any GPU call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# reference fwd+bwd FLOPs per sample per ELBO step (FlopCounterMode on the imported reference)
# (tests/golden/reference_flops.json, written by `python oracle/make_golden.py --flops`; COCO counted in round 4: 3 passes at
#  32x32 pixels and 102 caption steps, forward 937.75 MFLOP)
FLOP_PER_SAMPLE = {"multimnist": 427.24e6, "celeba": 1030.35e6, "coco": 2662.27e6}
ALGO_BYTES_PER_SAMPLE = {"multimnist": 0.59e9 / 256}          # 0.59 GB per step at B=256
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
DEFAULT_BATCH = {"multimnist": 256, "celeba": 512, "coco": 128}


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


def synthetic_batch_for(workload, B, seed):
    """config 3 (512,3,64,64) U[0,1) + (512,18) Bernoulli(0.3); config 5 (B,3,32,32) U[0,1) + (B,102,300) N(0,0.4^2)."""
    if workload == "multimnist":
        return synthetic_batch(B, seed)
    rng = np.random.default_rng(seed)
    if workload == "celeba":
        return (torch.from_numpy(rng.random((B, 3, 64, 64), dtype=np.float32)),
                torch.from_numpy((rng.random((B, 18)) < 0.3).astype(np.float32)))
    return (torch.from_numpy(rng.random((B, 3, 32, 32), dtype=np.float32)),
            torch.from_numpy((0.4 * rng.standard_normal((B, 102, 300))).astype(np.float32)))


def synthetic_sos():
    """Stand-in for the GloVe vector of '<s>' (coco/train.py feeds it as the first decoder input): fixed synthetic data."""
    return torch.from_numpy((0.4 * np.random.default_rng(4242).standard_normal(300)).astype(np.float32))


def measured_traffic(kernel_key):
    """HBM-side bytes per launch of a roofline kernel from rocprofv3 PMC passes (one pass per counter, no trace domains;
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH "HBM"): profiles/r03_traffic.json, written from the
    raw counter rows committed next to it.  None when the kernel has not been measured.  The newest round's file that holds
    the key wins (profiles/r04_traffic.json, then r03)."""
    for name in ("r04_traffic.json", "r03_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                e = json.load(f).get(kernel_key)
            if e:
                return e["bytes"], "profiles/%s: %s" % (name, e["source"])
        except Exception:
            pass
    return None, None


def host_cores():
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
    return int(os.environ.get("MMVAE_CPU_BASELINE_THREADS", min(cores, 64)))


def cpu_baseline(workload, B, D, a, b, budget_s=20.0):
    """The oracle (CPU restatement pinned to the reference) timed on this host's cores: reported, not the target."""
    from oracle import mmvae_ref as R
    cores = host_cores()
    torch.set_num_threads(cores)
    P = R.formula_params(workload, D, requires_grad=True)
    names = [n for n, _ in R.param_table(workload, D)]
    plist = [P[n] for n in names]
    m = [torch.zeros_like(p) for p in plist]
    v = [torch.zeros_like(p) for p in plist]
    sos = synthetic_sos() if workload == "coco" else None

    def step_losses():
        if workload == "multimnist":
            return R.multimnist_step_losses(P, a, b, True, 1e-3)[0]
        if workload == "celeba":
            return R.celeba_step_losses(P, a, b, True)[0]
        return R.coco_step_losses(P, a, b, sos, True, 1e-3)[0]

    times = []
    t_start = time.perf_counter()
    step = 0
    max_steps = 12 if workload == "multimnist" else 4
    while True:
        t0 = time.perf_counter()
        for p in plist:
            p.grad = None
        losses = step_losses()
        (losses[0] + losses[1] + losses[2]).backward()
        step += 1
        R.adam_step(plist, [p.grad for p in plist], m, v, step)
        dt = time.perf_counter() - t0
        if step > 1:
            times.append(dt)
        if (len(times) >= 2 and time.perf_counter() - t_start > budget_s) or len(times) >= max_steps:
            break
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "ELBO-steps/s", "cores": int(cores), "kind": "port",
            "sample": "%d full 3-pass fwd+bwd+Adam steps at B=%d after 1 warm-up (median), oracle/mmvae_ref.py (%s), fp32, torch %s CPU"
                      % (len(times), B, workload, torch.__version__),
            "samples_per_s": B / med}


def time_layer(eng, call, layer, iters=50):
    """Average launch time (us) of one layer's kernel, HIP events on the stream it is launched on, outside the step."""
    import ctypes
    s = torch.cuda.current_stream()
    st = ctypes.c_void_p(s.cuda_stream)
    eng.state.ensure_packed()
    call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer, 5, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    call("mmvae_mm_bench_layer", eng.h, eng.ws.data_ptr(), eng.ws.numel(), layer, iters, st)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def probe_steps(call, run, steps):
    """In-step kernel timing: `steps` steps with a HIP start/stop event pair on every launch that goes through the library's
    tagged launcher (include/mmvae_hip.h mmvae_debug_probe).  -> [{tag, kernel, family, launches_per_step, us, flops}] per
    distinct (tag, kernel), `us` the mean duration of one launch inside the running step."""
    import ctypes
    run()
    torch.cuda.synchronize()
    call("mmvae_debug_probe", 1)
    for _ in range(steps):
        run()
    call("mmvae_debug_probe", 0)
    torch.cuda.synchronize()
    cap = 1 << 22
    buf = ctypes.create_string_buffer(cap)
    call("mmvae_debug_probe_read", buf, cap)
    acc = {}
    for line in buf.value.decode().splitlines():
        tag, kernel, us, flops = line.split("\t")
        kernel = kernel.strip("() ")
        e = acc.setdefault((tag, kernel), {"tag": tag, "kernel": kernel, "family": kernel_family(kernel), "n": 0, "us": 0.0, "flops": float(flops)})
        e["n"] += 1
        e["us"] += float(us)
    out = []
    for e in acc.values():
        out.append({"tag": e["tag"], "kernel": e["kernel"], "family": e["family"], "launches_per_step": e["n"] / steps,
                    "us": e["us"] / e["n"], "flops": e["flops"]})
    return out


WGRAD_FAMILY = "weight gradient (wgrad_ring_kernel, wgrad_kernel, wgrad_multi_kernel + their reduce launches)"


def kernel_family(kernel):
    """Kernels that do ONE job together are one family: every weight-gradient GEMM kernel and the reduce launches that sum
    their partial copies (the reduce launches carry time and no FLOPs -- a family that left them out would look better than
    it is)."""
    base = kernel.split("<")[0].strip()
    if base.startswith("wgrad_"):
        return WGRAD_FAMILY
    return base


def family_roofline(rows, family):
    """Sum of algorithmic FLOPs / sum of in-step launch time over ALL of one step's launches of `family` (launches without
    FLOPs -- reduce kernels -- count with their time)."""
    sel = [r for r in rows if r["family"] == family]
    us = sum(r["us"] * r["launches_per_step"] for r in sel)
    fl = sum(r["flops"] * r["launches_per_step"] for r in sel)
    n = sum(r["launches_per_step"] for r in sel)
    ach = fl / (us * 1e-6) / 1e12 if us > 0 else 0.0
    return {"kernel": "%s: %d launches per step, timed inside the step" % (family, round(n)), "bound": "mfma", "achieved": ach,
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "us_per_launch": us / max(n, 1),
            "flops_per_launch": fl / max(n, 1), "us_per_step": us,
            "launches": [{"tag": r["tag"] if r["flops"] > 0 else r["kernel"].split("<")[0], "n": round(r["launches_per_step"], 2), "us": round(r["us"], 2),
                          "tflops": round(r["flops"] / (r["us"] * 1e-6) / 1e12, 1)}
                         for r in sorted(sel, key=lambda r: -r["us"] * r["launches_per_step"])]}


def kernels_in_step(rows, top=16):
    tot = {}
    for r in rows:
        e = tot.setdefault(r["kernel"] if r["family"] not in ("convres_kernel", WGRAD_FAMILY) else r["family"], {"launches_per_step": 0.0, "us_per_step": 0.0, "flops": 0.0})
        e["launches_per_step"] += r["launches_per_step"]
        e["us_per_step"] += r["us"] * r["launches_per_step"]
        e["flops"] += r["flops"] * r["launches_per_step"]
    out = []
    for k, e in sorted(tot.items(), key=lambda kv: -kv[1]["us_per_step"])[:top]:
        out.append({"kernel": k, "launches_per_step": round(e["launches_per_step"], 2), "us_per_step": round(e["us_per_step"], 1),
                    "tflops": round(e["flops"] / (e["us_per_step"] * 1e-6) / 1e12, 1) if e["flops"] > 0 else None})
    return out


# MultiMNIST layers that run on convres_kernel (multimodal-vae_amd/csrc/convres.hip), by their mmvae_mm_bench_layer names
MM_CONVRES_LAYERS = ["enc_conv2", "enc_conv3", "enc_conv4", "dec_convT1", "dec_convT2", "dec_convT3",
                     "enc_conv2_dgrad", "enc_conv3_dgrad", "enc_conv4_dgrad", "dec_convT1_dgrad", "dec_convT2_dgrad", "dec_convT3_dgrad"]


MM_WGRAD_LAYERS = ["dec_convT1_wgrad", "dec_convT2_wgrad", "dec_convT3_wgrad", "enc_conv2_wgrad", "enc_conv3_wgrad", "enc_conv4_wgrad"]


def roofline_entry(eng, call, layer, kernel, B, D):
    lb = layer.encode()
    us = time_layer(eng, call, lb)
    algo = call("mmvae_mm_layer_algo_flops", eng.h, lb)
    executed = call("mmvae_mm_layer_flops", eng.h, lb)
    algo_bytes = call("mmvae_mm_layer_algo_bytes", eng.h, lb)
    traffic, src = measured_traffic("%s:%d:%d" % (layer, B, D))
    ach = algo / (us * 1e-6) / 1e12
    e = {"kernel": "%s (%s)" % (kernel, layer), "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
         "frac": ach / PEAK_BF16_TFLOPS, "us_per_launch": us, "flops_per_launch": algo, "executed_flops_per_launch": executed,
         "algorithmic_bytes_per_launch": algo_bytes, "hbm_frac_algorithmic": algo_bytes / (us * 1e-6) / (PEAK_HBM_GBS * 1e9),
         "traffic": traffic, "traffic_source": src}
    if traffic:
        e["hbm_frac_measured"] = traffic / (us * 1e-6) / (PEAK_HBM_GBS * 1e9)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="multimnist", choices=("multimnist", "celeba", "coco"))
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (default: the workload's named batch)")
    ap.add_argument("--n_latents", type=int, default=100)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (1 GPU only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-overlap", action="store_true", help="N>1, MultiMNIST: all-reduce the decoders' gradients while the encoders' backward runs")
    ap.add_argument("--knob", action="append", default=[], help="name=value: an A/B switch of the library (mmvae_debug_set), measurement aid")
    ap.add_argument("--loader", action="store_true",
                    help="also time the step fed by data.DeviceBatcher (pinned uint8 pixels + second modality, double-buffered async H2D, "
                         "u8->f32 on the device) inside the timed loop: reported as `with_loader` next to the resident-input `value`")
    ap.add_argument("--no-probe", action="store_true", help="skip the in-step kernel timing (roofline falls back to the whole step)")
    args = ap.parse_args()
    wl = args.workload

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd import core
    from multimodal_vae_amd.init import default_init_
    from multimodal_vae_amd._lib import call

    for kv in args.knob:
        k, v = kv.split("=")
        call("mmvae_debug_set", k.encode(), int(v))

    from multimodal_vae_amd import dp
    all_reduce = None
    if world > 1:
        dp.init_distributed("nccl", dev)                   # backend "nccl" IS RCCL on ROCm
        # SUM all-reduce of the flat gradient, 1/world folded into Adam.  --dp-overlap (MultiMNIST): the decoders'
        # gradient ranges go out while the encoders' backward still runs (core.FusedELBOStep._call_dp_overlap).  Not the
        # default: measured on one GPU with a world-1 RCCL group the extra communication stream (parked on the early-gradient
        # event) lands on a hardware queue shared with the step's streams and stalls them (0.91 -> 2.2 ms per step), and even
        # without that the split costs 0.05 ms against the ~0.06 ms of a 9.4 MB exchange it can hide
        all_reduce = dp.GradAllReduce(overlap=args.dp_overlap)

    B, D = args.batch or DEFAULT_BATCH[wl], args.n_latents
    seed = dp.rank_seed(1234, rank)
    a, b = synthetic_batch_for(wl, B, seed)
    if wl == "multimnist":
        state = core.MultimnistState(D, dev)
        default_init_(state, seed=1234)                   # identical replicas on every rank
        eng = core.FusedELBOStep(state, B, lr=1e-3, seed=seed, world_size=world, all_reduce=all_reduce)
        metric = "ELBO-steps/sec (whole node), MultiMNIST b=%d per GPU" % B
        workload = "multimnist_50x50_conv_mmvae_gru_text_3pass_elbo_step"
        dtype = "bf16"
    elif wl == "celeba":
        state = core.CelebaState(D, dev)
        default_init_(state, seed=1234)
        eng = core.FusedCelebaStep(state, B, seed=seed, world_size=world, all_reduce=all_reduce)
        metric = "ELBO-steps/sec (whole node), CelebA 64x64 b=%d per GPU" % B
        workload = "celeba_64x64_conv_mmvae_18_attributes_3pass_elbo_step"
        dtype = "bf16"
    else:
        state = core.CocoState(D, dev)
        default_init_(state, seed=1234)
        eng = core.FusedCocoStep(state, B, synthetic_sos(), seed=seed, world_size=world, all_reduce=all_reduce)
        metric = "ELBO-steps/sec (whole node), COCO 32x32 + 102-step captions b=%d per GPU" % B
        workload = "coco_32x32_conv_mmvae_glove_caption_gru_3pass_elbo_step"
        dtype = "bf16 (MFMA operands of the image half and of the caption GRUs; f32 accumulation, gate math, state; MMVAE_COCO_TEXT_FP32=1: f32 caption GRUs)"
    dp.broadcast_flat(state.params)
    dp.broadcast_flat(state.bn_stats)
    a_d, b_d = a.to(dev).contiguous(), b.to(dev).contiguous()

    use_graph = args.graph and world == 1
    if use_graph:
        eng.capture(a_d, b_d)
        run = eng.replay
    else:
        run = lambda: eng(a_d, b_d)                      # noqa: E731

    def barrier():
        dp.barrier(dev)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    call("mmvae_debug_flops", 1)
    eng(a_d, b_d)                                         # (eagerly, also with --graph: a replay does not pass through the launchers)
    executed = call("mmvae_debug_flops", 1)               # GEMM FLOPs one step enqueues (host-side count)
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
        "metric": metric,
        "value": steps_per_s,              # every rank runs one step per wall-clock step: whole-node steps/s of B*world samples
        "unit": "ELBO-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "samples_per_s": steps_per_s * B * world,
        "config": {"workload": workload, "batch_per_gpu": B, "global_batch": B * world, "n_latents": D,
                   "parallelism": "dp%d" % world, "hip_graph": use_graph, "optimizer": "adam",
                   "mfma": "bf16 in / fp32 acc; fp32 master weights, BN stats, PoE/KL/BCE/NLL, Adam"},
        "final_losses": losses,
        "executed_flops_per_step": executed,
        "step_mfma_frac_executed": steps_per_s * executed / (PEAK_BF16_TFLOPS * 1e12),
    }
    if FLOP_PER_SAMPLE[wl]:
        result["reference_flops_per_step"] = B * FLOP_PER_SAMPLE[wl]
        result["step_mfma_frac"] = steps_per_s * B * FLOP_PER_SAMPLE[wl] / (PEAK_BF16_TFLOPS * 1e12)
    if wl in ALGO_BYTES_PER_SAMPLE:
        result["algorithmic_bytes_per_step"] = B * ALGO_BYTES_PER_SAMPLE[wl]
        result["step_hbm_frac"] = steps_per_s * B * ALGO_BYTES_PER_SAMPLE[wl] / (PEAK_HBM_GBS * 1e9)

    # ---- the same steps fed by the asynchronous loader ("async H2D image + caption pipeline"): a synthetic
    # dataset of 8 batches on the host, gathered / copied / converted batch by batch while the previous step runs
    if args.loader:
        from multimodal_vae_amd.data import DeviceBatcher
        nb_ds = int(os.environ.get("MMVAE_BENCH_DATASET_BATCHES", "8"))
        rng = np.random.default_rng(seed + 99)
        a_h, b_h = synthetic_batch_for(wl, nb_ds * B, seed + 99)
        img_u8 = (a_h * 255.0).round().clamp(0, 255).to(torch.uint8)
        if wl == "multimnist":
            img_u8 = img_u8[:, 0]
        loader = DeviceBatcher(img_u8, b_h, B, dev, shuffle=True, seed=seed)

        def loader_steps(n):
            done = 0
            while done < n:
                for im, tx in loader:
                    eng(im, tx)
                    done += 1
                    if done == n:
                        break
        loader_steps(max(100, args.warmup))          # (the loader-fed loop has a transient of its own -- first touches of the pinned dataset over
        #                                              the host link, the worker thread's start: the first ~200 steps run 5-8 % slower)
        barrier()
        gc.collect()
        gc.disable()        # as in the resident-input loop above: a generational collection in the enqueue thread stalls the GPU for milliseconds
        t0 = time.perf_counter()
        loader_steps(args.steps)
        t_host = time.perf_counter() - t0          # the enqueue thread's loop alone: equal to the wall time = the loop is host-bound
        barrier()
        dtl = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dtl], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtl = float(t.item())
        result["with_loader"] = {"value": args.steps / dtl, "unit": "ELBO-steps/s", "ms_per_step": 1e3 * dtl / args.steps,
                                 "vs_resident": (args.steps / dtl) / steps_per_s,
                                 "enqueue_loop_ms_per_step": 1e3 * t_host / args.steps,
                                 "h2d_bytes_per_step": int(img_u8[0].numel() * B + b_h[0].numel() * b_h.element_size() * B),
                                 "loader": "data.DeviceBatcher: %d-batch host dataset, gather into pinned staging on a worker thread, async H2D on a "
                                           "copy stream, u8->f32 on the device, four slots, slot reuse paced on the worker thread" % nb_ds}

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

    # in-step kernel timing: EVERY rank runs the probed steps (they contain the gradient exchange), rank 0 reports
    rows = [] if (args.no_probe or use_graph) else probe_steps(call, run, 20)
    if rank == 0:
        if rows:
            result["kernels_in_step"] = kernels_in_step(rows)
        fam_us, fam_fl = {}, {}
        for r in rows:
            fam_us[r["family"]] = fam_us.get(r["family"], 0.0) + r["us"] * r["launches_per_step"]
            fam_fl[r["family"]] = fam_fl.get(r["family"], 0.0) + r["flops"] * r["launches_per_step"]
        fam_us = {k: v for k, v in fam_us.items() if fam_fl[k] > 0}           # GEMM families only
        if fam_us:
            # ---- roofline: the GEMM kernel FAMILY with the most GPU time in the step -- all of its launches, reduce launches
            # included -- timed in the step; the runner-up family in `roofline_second`
            order = sorted(fam_us, key=fam_us.get, reverse=True)
            for slot, fam in zip(("roofline", "roofline_second"), order[:2]):
                e = family_roofline(rows, fam)
                e["traffic"] = None
                if wl == "multimnist" and fam == "convres_kernel":
                    iso_us = sum(time_layer(eng, call, l.encode()) for l in MM_CONVRES_LAYERS)
                    iso_fl = sum(call("mmvae_mm_layer_algo_flops", eng.h, l.encode()) for l in MM_CONVRES_LAYERS)
                    e["frac_isolated"] = iso_fl / (iso_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS
                    e["us_per_step_isolated"] = iso_us
                    # HBM-side traffic of the family's longest launch (the forward of the last 64->32 ConvTranspose2d)
                    e["traffic"], e["traffic_source"] = measured_traffic("dec_convT3:%d:%d" % (B, D))
                    e["traffic_launch"] = "dec_convT3"
                    e["algorithmic_bytes_of_that_launch"] = call("mmvae_mm_layer_algo_bytes", eng.h, b"dec_convT3")
                if wl == "multimnist" and fam == WGRAD_FAMILY:
                    # the conv layers' weight gradients replayed alone (kernel + reduce, as in the step; the Linear / GRU ones have
                    # no replay hook) and the PMC traffic of the family's longest launch
                    iso_us = sum(time_layer(eng, call, l.encode()) for l in MM_WGRAD_LAYERS)
                    iso_fl = sum(call("mmvae_mm_layer_algo_flops", eng.h, l.encode()) for l in MM_WGRAD_LAYERS)
                    e["frac_isolated_conv_layers"] = iso_fl / (iso_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS
                    e["us_isolated_conv_layers"] = iso_us
                    e["traffic"], e["traffic_source"] = measured_traffic("dec_convT3_wgrad:%d:%d" % (B, D))
                    e["traffic_launch"] = "dec_convT3_wgrad (kernel + reduce)"
                    e["algorithmic_bytes_of_that_launch"] = call("mmvae_mm_layer_algo_bytes", eng.h, b"dec_convT3_wgrad")
                if wl == "celeba":
                    key = {WGRAD_FAMILY: "ca_wgrad", "gemm_gather_kernel": "ca_gemm_gather", "convres_kernel": "ca_convres"}.get(fam)
                    if key:
                        e["traffic"], e["traffic_source"] = measured_traffic("%s:%d:%d" % (key, B, D))
                        e["traffic_launch"] = "every launch of the family in one step (rocprofv3 --pmc over bench.py --workload celeba)"
                result[slot] = e
        if wl == "multimnist":
            result["roofline_fwd_isolated"] = roofline_entry(eng, call, "dec_convT3", "convres_kernel", B, D)
            result["roofline_wgrad"] = roofline_entry(eng, call, "dec_convT3_wgrad", "wgrad_ring_kernel + wgrad_ring_reduce_kernel", B, D)
            result["roofline_dgrad"] = roofline_entry(eng, call, "dec_convT3_dgrad", "convres_kernel", B, D)
        if "roofline" not in result:
            # no probe: the whole step against the MFMA roof on the executed GEMM FLOPs
            ms = result["ms_per_step"]
            ach = executed / (ms * 1e-3) / 1e12
            result["roofline"] = {"kernel": "whole 3-pass step (all GEMM launches)", "bound": "mfma", "achieved": ach,
                                  "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                                  "us_per_launch": ms * 1e3, "flops_per_launch": executed, "traffic": None}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wl, B, D, a, b)
            result["speedup_vs_cpu_baseline"] = steps_per_s / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
