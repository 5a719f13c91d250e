"""GPU: the data-parallel exchange step on its real backend.  ``init_process_group("nccl")`` IS RCCL on ROCm
(what /root/reference/src/utils/create_trainer.py:50-58 asks DDP for); a one-GPU box can run it at world size 1:
the gradient arena goes through ``ncclAllReduce`` and the BatchNorm buffer block through ``ncclBroadcast`` every
step.  Two training steps with the group must end bit-identical to the same two steps without one."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from tests.util import blob_cloud

pytestmark = pytest.mark.gpu
HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _trainer():
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.train import Trainer
    cfg = Config(encoder=Encoder(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1))
    return Trainer(cfg=cfg, image_size=[1, 32, 32, 32], output_shape=HEADS, device="cuda", seed=5)


def _two_steps(tr):
    out = []
    torch.manual_seed(99)   # the heads hold nn.Dropout: both runs must draw the same masks
    for seed, n in ((21, 400), (22, 330)):
        coords, feats = blob_cloud(seed, 3, [32, 32, 32], n, 1)
        g = torch.Generator().manual_seed(seed)
        labels = {k: torch.randint(0, v, (3,), generator=g).numpy() for k, v in HEADS.items()}
        loss, _ = tr.train_step(*tr.stage((coords, feats, 3), labels))
        out.append(float(loss))
    torch.cuda.synchronize()
    return out


def test_rccl_world_size_one_matches_no_process_group(monkeypatch):
    assert not dist.is_initialized()
    plain = _trainer()
    losses_plain = _two_steps(plain)

    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    from sparseeventid_amd.distributed import init_from_env
    calls = {"all_reduce": 0, "broadcast": 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)

    def count_bc(*a, **k):
        calls["broadcast"] += 1
        return real_bc(*a, **k)

    try:
        rank, world, _ = init_from_env("nccl", force=True)
        assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
        monkeypatch.setattr(dist, "all_reduce", count_ar)
        monkeypatch.setattr(dist, "broadcast", count_bc)
        grouped = _trainer()
        grouped.arena.broadcast_parameters()
        assert torch.equal(grouped.arena.flat, _trainer().arena.flat)
        losses_grouped = _two_steps(grouped)
        # the exchange really ran: one all-reduce of the gradient arena and one buffer broadcast per step
        assert calls["all_reduce"] == 2 and calls["broadcast"] == 1 + 2
        # a collective of its own, checked against its definition (sum over one rank, scaled by 1/world)
        g = torch.arange(1000, device="cuda", dtype=torch.float32)
        grouped.arena.grad[:1000].copy_(g)
        grouped.arena.all_reduce_mean()
        assert torch.equal(grouped.arena.grad[:1000], g)
        dist.barrier()
    finally:
        monkeypatch.undo()
        if dist.is_initialized():
            dist.destroy_process_group()
    assert losses_plain == losses_grouped
    assert torch.equal(plain.arena.flat, grouped.arena.flat)
    for (k, a), (_, b) in zip(plain.encoder.named_buffers(), grouped.encoder.named_buffers()):
        assert torch.equal(a, b), k
