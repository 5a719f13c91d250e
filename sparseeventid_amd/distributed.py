"""Data-parallel glue: one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on
ROCm).  Replaces the reference's Lightning-DDP / horovod paths
(/root/reference/src/utils/create_trainer.py:46-58; legacy src/utils/torch/distributed_trainer.py:34-97).

Events shard as independent batch items; BatchNorm statistics stay per-rank (the reference uses no
SyncBN).  The only exchange is ONE all-reduce(sum) of the flat fp32 gradient arena
(20.9 M floats = 83.5 MB for the default network) followed by a 1/world scale (DDP mean semantics).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process gets no group unless `force` is set: with a group, every
    collective of the data-parallel path (gradient all-reduce, parameter / buffer broadcast) really runs through the
    backend at world size 1 -- how the RCCL path is exercised on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatArena:
    """All parameters (and all gradients) of a set of modules as views into ONE contiguous fp32
    buffer each, so the gradient exchange is a single collective and the optimizer a single kernel."""

    def __init__(self, modules):
        params = []
        seen = set()
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        self.params = params
        # every parameter starts on a 256-byte boundary: the conv kernels read weights / biases with 16-byte loads
        align = 64
        total = sum((p.numel() + align - 1) // align * align for p in params)
        dev = params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._end = {}
        self._views = []
        self.spans = []                                     # (offset, numel) of every parameter in the arena
        for p in params:
            n = p.numel()
            self.spans.append((off, n))
            self._end[id(p)] = off + (n + align - 1) // align * align
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            self._views.append(p.grad)
            off += (n + align - 1) // align * align
        self.numel = total                                  # arena length (parameters + alignment padding)
        self.n_params = sum(p.numel() for p in params)

    def zero_grad(self):
        self.grad.zero_()

    def detach_grads(self):
        """Before backward(): every parameter's .grad -> None, so that autograd ASSIGNS the gradient tensors the backward
        functions return instead of launching one `grad += g` kernel per parameter (237 for the default network)."""
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """After backward(): the per-parameter gradient tensors -> their slots of the flat gradient arena with ONE
        multi-tensor copy, and .grad points at the arena views again.  Parameters that received no gradient keep the
        zeros zero_grad() left in their slot.  Equal to accumulation into a zeroed arena (g + 0 == g)."""
        dst, src = [], []
        for p, v in zip(self.params, self._views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g.reshape(v.shape) if g.shape != v.shape else g)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def offset_after(self, module):
        """Arena offset just past the parameters of `module` (they must be a leading run of the arena)."""
        ids = {id(p) for p in module.parameters()}
        n = len(ids)
        if {id(p) for p in self.params[:n]} != ids:
            raise ValueError("module parameters are not a leading run of the arena")
        return self._end[id(self.params[n - 1])] if n else 0

    def all_reduce_mean(self, start=0):
        """Sum gradients over ranks (RCCL all-reduce over xGMI), then scale by 1/world.  `start` > 0 skips the
        leading (frozen) part of the arena."""
        if dist.is_initialized():   # also at world size 1: the collective runs (x * 1.0 is exact)
            g = self.grad[start:] if start else self.grad
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            g.mul_(1.0 / dist.get_world_size())

    def broadcast_parameters(self, src=0):
        if dist.is_initialized():
            dist.broadcast(self.flat, src=src)


class FlatBuffers:
    """All floating-point buffers (BatchNorm running statistics) of a set of modules as views into ONE contiguous
    tensor, so that what DDP's ``broadcast_buffers=True`` does before every forward (rank 0's buffers to every
    rank; torch default, create_trainer.py:50-58 builds DDP with it) is a single small broadcast."""

    def __init__(self, modules):
        bufs, seen = [], set()
        for m in modules:
            for b in m.buffers():
                if b.dtype == torch.float32 and id(b) not in seen:
                    seen.add(id(b))
                    bufs.append(b)
        self.bufs = bufs
        self.flat = None
        if bufs:
            self.flat = torch.zeros(sum(b.numel() for b in bufs), dtype=torch.float32, device=bufs[0].device)
            off = 0
            for b in bufs:
                n = b.numel()
                self.flat[off:off + n].copy_(b.data.reshape(-1))
                b.data = self.flat[off:off + n].view(b.shape)
                off += n

    def broadcast(self, src=0):
        if self.flat is not None and dist.is_initialized():
            dist.broadcast(self.flat, src=src)


def broadcast_buffers(modules, src=0):
    """What DDP's broadcast_buffers does for the BN running statistics."""
    if not dist.is_initialized():
        return
    for m in modules:
        for b in m.buffers():
            dist.broadcast(b, src=src)
