"""Single-node data parallelism for the training step: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" runs the same bookkeeping on CPU for tests).

The reference is single-process (its nn.DataParallel wrapper is bypassed), so nothing here is a
translation.  Partition: the global batch is cut rank-major into contiguous slices; each rank encodes its slice,
the L2-normalised embeddings are all-gathered (one fused [Bl, 2E] message per rank), every rank evaluates the
identical global-batch loss head and keeps rows [r*Bl, (r+1)*Bl) of dL/dI and dL/dT.  Parameter gradients are
all-reduced with SUM (the loss already carries its 1/B factors) bucket by bucket as the backward produces them,
on RCCL's own stream, and joined before the optimiser step.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  No-op at world size 1."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # SC_DIST_BACKEND=gloo rehearses the N > 1 bookkeeping with several ranks on ONE GPU (RCCL refuses duplicate devices)
            backend = os.environ.get("SC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_bounds(global_batch: int, rank: int, world: int):
    """Rank-major contiguous slice of the global batch."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by the {world} ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def all_gather_embeddings(img_local: torch.Tensor, txt_local: torch.Tensor):
    """[Bl,E] x 2 -> [B,E] x 2 on every rank, rank-major (one collective on the fused [Bl, 2E] buffer)."""
    if world_size() == 1:
        return img_local, txt_local
    bl, e = img_local.shape
    fused = torch.cat([img_local, txt_local], dim=1).contiguous()
    out = torch.empty(bl * world_size(), 2 * e, dtype=fused.dtype, device=fused.device)
    if dist.get_backend() == "gloo" and fused.is_cuda:   # rehearsal path only: gloo has no CUDA all_gather_into_tensor
        parts = [torch.empty_like(fused) for _ in range(world_size())]
        dist.all_gather(parts, fused)
        out = torch.cat(parts, dim=0)
    else:
        dist.all_gather_into_tensor(out, fused)
    return out[:, :e].contiguous(), out[:, e:].contiguous()


def local_rows(full: torch.Tensor):
    """This rank's rows of a gathered [B,E] tensor (the backward of the gather needs no collective)."""
    if world_size() == 1:
        return full
    a, b = shard_bounds(full.shape[0], get_rank(), world_size())
    return full[a:b].contiguous()


def broadcast_parameters(model):
    if world_size() > 1:
        dist.broadcast(model.flat, src=0)
        model.refresh_shadows(full=True)


class GradSync:
    """Bucketed SUM all-reduce of model.flat_grad, launched as each bucket becomes final during the backward."""

    def __init__(self, model):
        self.model = model
        self.spans = dict(model.buckets)
        self.pending = []
        model.comm = self

    def bucket_ready(self, name: str):
        if world_size() == 1:
            return
        a, b = self.spans[name]
        self.pending.append(dist.all_reduce(self.model.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def wait_all(self):
        for w in self.pending:
            w.wait()
        self.pending = []
