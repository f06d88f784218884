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
