"""CPU: the N>1 path with world_size-2 gloo -- flat gradient arena all-reduce (mean), parameter
broadcast, buffer broadcast, equality of replicas after an optimizer step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from sparseeventid_amd.distributed import FlatArena, FlatBuffers, broadcast_buffers, init_from_env
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # replicas start DIFFERENT on purpose
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    arena = FlatArena([net])
    assert arena.n_params == sum(p.numel() for p in net.parameters()) and arena.numel >= arena.n_params
    arena.broadcast_parameters(0)
    broadcast_buffers([net], 0)
    p0 = arena.flat.clone()
    # parameters are views of the arena: an update through the arena is seen by the module
    opt = torch.optim.SGD([torch.nn.Parameter(arena.flat)], lr=0.1)
    opt.param_groups[0]["params"][0].grad = arena.grad
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(rank))   # each rank its own shard
    arena.zero_grad()
    net(x).square().mean().backward()
    local = arena.grad.clone()
    arena.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(arena.grad, sum(gathered) / world, atol=1e-7)
    opt.step()
    assert torch.allclose(net[0].weight.reshape(-1), arena.flat[:30])
    flats = [torch.zeros_like(arena.flat) for _ in range(world)]
    dist.all_gather(flats, arena.flat)
    assert torch.equal(flats[0], flats[1])             # replicas stay identical
    assert not torch.equal(arena.flat, p0)
    # DDP broadcast_buffers semantics through ONE flat broadcast: the ranks' BN running statistics have diverged
    # (different shards); after the broadcast every rank holds rank 0's
    fb = FlatBuffers([net])
    assert net[1].running_mean.data_ptr() == fb.flat.data_ptr()
    mine = net[1].running_var.clone()
    fb.broadcast(0)
    rv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rv, net[1].running_var)
    assert torch.equal(rv[0], rv[1]) and (rank == 0) == torch.equal(mine, net[1].running_var)
    out.put((rank, float(arena.flat.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert abs(res[0] - res[1]) < 1e-6


def _solo_worker(port, out):
    """world size 1 with a forced group: every collective of the exchange step still goes through the backend."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from sparseeventid_amd.distributed import FlatArena, FlatBuffers, init_from_env
    assert init_from_env("gloo")[:2] == (0, 1) and not dist.is_initialized()      # no group unless forced
    assert init_from_env("gloo", force=True)[:2] == (0, 1) and dist.is_initialized()
    calls = []
    real_ar, real_bc = dist.all_reduce, dist.broadcast
    dist.all_reduce = lambda *a, **k: (calls.append("ar"), real_ar(*a, **k))[1]
    dist.broadcast = lambda *a, **k: (calls.append("bc"), real_bc(*a, **k))[1]
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    arena, fb = FlatArena([net]), FlatBuffers([net])
    arena.broadcast_parameters(0)
    fb.broadcast(0)
    arena.zero_grad()
    net(torch.randn(5, 4)).sum().backward()
    g = arena.grad.clone()
    arena.all_reduce_mean()
    assert torch.equal(g, arena.grad) and calls == ["bc", "bc", "ar"]
    out.put("ok")
    dist.destroy_process_group()


def test_single_rank_forced_group_runs_the_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and q.get(timeout=5) == "ok"
