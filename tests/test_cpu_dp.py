"""CPU (gloo, world_size 2): the data-parallel exchange of the fused step -- one SUM all-reduce of the flat gradient
buffer, 1/world folded into Adam -- gives every rank the Adam update of the mean gradient, and replicas stay equal."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import mmvae_ref as R


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd import dp
    r, w, _ = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    n = 4099
    g = torch.Generator().manual_seed(dp.rank_seed(1234, rank))
    params = torch.randn(n, generator=torch.Generator().manual_seed(5 + rank))     # deliberately different replicas
    dp.broadcast_flat(params, src=0)                                               # -> identical
    grads = torch.randn(n, generator=g)                                            # rank-local "gradient"
    local = grads.clone()
    dp.GradAllReduce(bucket_bytes=4096)(grads)                                     # bucketed SUM over ranks
    # what the engine's Adam does with grad_scale = 1/world (checked against the oracle's Adam)
    m, v = torch.zeros(n), torch.zeros(n)
    R.adam_step([params], [grads / world], [m], [v], 1)
    torch.save({"params": params, "local": local, "sum": grads}, os.path.join(out_dir, f"r{rank}.pt"))
    dp.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_world2_allreduce_mean_and_replicas(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"r{r}.pt") for r in range(2))
    np.testing.assert_allclose(a["sum"].numpy(), (a["local"] + b["local"]).numpy(), rtol=1e-6)
    np.testing.assert_array_equal(a["sum"].numpy(), b["sum"].numpy())
    np.testing.assert_array_equal(a["params"].numpy(), b["params"].numpy())       # replicas identical after the step
    assert not np.array_equal(a["local"].numpy(), b["local"].numpy())              # shards really differed


def test_single_process_is_a_noop():
    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd import dp
    t = torch.arange(8.0)
    dp.GradAllReduce()(t)
    dp.broadcast_flat(t)
    dp.barrier()
    assert t.tolist() == list(range(8))
    assert dp.rank_seed(1234, 3) == 1237


# ---------------------------------------------------------------------------------------------------------------------
# The engine's own optimizer_step under gloo: the C-ABI Adam entry point is replaced by the oracle's Adam on the CPU
# tensors it is handed (there is no GPU here); everything else -- _dp_active, the order all-reduce -> Adam, the 1/world
# grad_scale argument, the deferred weight pack -- is the product code of core._FusedStepBase.
def _step_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import multimodal_vae_amd  # noqa: F401
    from multimodal_vae_amd import core, dp
    dp.init_distributed("gloo")
    n = 1031
    events = []

    class State:                                   # the slice of PlanState optimizer_step touches
        def __init__(self):
            self.params = torch.randn(n, generator=torch.Generator().manual_seed(77))      # replicas start equal
            self.grads = torch.randn(n, generator=torch.Generator().manual_seed(dp.rank_seed(1234, rank)))
            self.nparams = n
            self.pack_pending = False

        def pack_weights(self):
            events.append("pack")

    class Collective(dp.GradAllReduce):
        def __call__(self, flat):
            events.append("all_reduce")
            super().__call__(flat)

    tensors = {}

    def fake_ptr(t):
        tensors[id(t)] = t
        return t

    def fake_call(name, *a):
        assert name == "mmvae_adam_step", name
        events.append("adam")
        params, grads, m, v, count, _state, lr, b1, b2, eps, scale, _stream = a
        assert count == n and abs(lr - 1e-3) < 1e-12 and (b1, b2) == (0.9, 0.999)
        tensors["scale"] = scale
        tensors["grads_seen_by_adam"] = grads.clone()
        R.adam_step([params], [grads * scale], [m], [v], 1)
        return 0

    core.call, core.ptr, core._stream = fake_call, fake_ptr, lambda: None
    eng = core._FusedStepBase.__new__(core._FusedStepBase)
    eng.state = State()
    eng.world_size, eng.all_reduce = world, Collective(bucket_bytes=1024)
    eng.lr, eng.betas, eng.eps = 1e-3, (0.9, 0.999), 1e-8
    eng.exp_avg, eng.exp_avg_sq = torch.zeros(n), torch.zeros(n)
    eng.adam_state = torch.zeros(2, dtype=torch.int64)
    local = eng.state.grads.clone()
    assert eng._dp_active()
    eng.optimizer_step()
    torch.save({"params": eng.state.params, "local": local, "seen": tensors["grads_seen_by_adam"], "scale": tensors["scale"],
                "events": events}, os.path.join(out_dir, f"s{rank}.pt"))
    dp.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_world2_optimizer_step_orders_allreduce_before_adam(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_step_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"s{r}.pt") for r in range(2))
    for r in (a, b):
        assert r["events"] == ["all_reduce", "adam", "pack"]                      # exchange first, then the update, then the re-pack
        assert r["scale"] == 0.5                                                   # 1/world folded into Adam
        np.testing.assert_allclose(r["seen"].numpy(), (a["local"] + b["local"]).numpy(), rtol=1e-6)   # Adam saw the SUM
    np.testing.assert_array_equal(a["params"].numpy(), b["params"].numpy())       # replicas identical after the step
    # ... and equal to Adam on the MEAN gradient of the two shards
    p = torch.randn(1031, generator=torch.Generator().manual_seed(77))
    R.adam_step([p], [(a["local"] + b["local"]) * 0.5], [torch.zeros(1031)], [torch.zeros(1031)], 1)
    np.testing.assert_allclose(a["params"].numpy(), p.numpy(), rtol=1e-6, atol=1e-7)
