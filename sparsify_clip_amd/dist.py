"""Single-node data parallelism for the training step: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" runs the same bookkeeping on CPU for tests).

The reference is single-process (its nn.DataParallel wrapper is bypassed), so nothing here is a
translation.  Partition: the global batch is cut rank-major into contiguous slices; each rank encodes its slice,
the L2-normalised embeddings are all-gathered (one fused [2, Bl, E] message per rank); the global-batch loss head is evaluated
by rows: rank r runs the O(B^2) sweeps over rows [r*Bl, (r+1)*Bl) against every column, the ranks exchange their LSE statistics
once (exchange_packets), and each rank ends with the same loss value and its own rows of dL/dI and dL/dT (shapes the row-block
kernels do not take fall back to every rank evaluating the whole loss head and keeping its rows).  Parameter gradients are
all-reduced with SUM (the loss already carries its 1/B factors) bucket by bucket as the backward produces them,
on RCCL's own stream, and joined before the optimiser step.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None, force=False):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  No-op at world size 1 unless
    `force` (a world-1 group: every collective below then really goes through the backend - how the RCCL path is exercised on
    a one-GPU box, tests/test_gpu_dp.py)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
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


def active():
    """True when a process group exists: the collectives run (also at world size 1, see init_process_group(force=True))."""
    return dist.is_initialized()


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_bounds(global_batch: int, rank: int, world: int):
    """Rank-major contiguous slice of the global batch."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by the {world} ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


_gather_bufs = {}


def gather_send_buffer(bl: int, e: int, device):
    """The rank's fused send buffer [2, Bl, E] (image rows, then text rows): the L2-normalisation kernels write their outputs
    straight into its two halves, so the exchange needs no torch.cat."""
    key = (bl, e, str(device), world_size())
    buf = _gather_bufs.get(key)
    if buf is None:
        buf = _gather_bufs[key] = (torch.empty(2, bl, e, dtype=torch.float32, device=device),
                                   torch.empty(world_size(), 2, bl, e, dtype=torch.float32, device=device))
    return buf


def all_gather_embeddings(img_local: torch.Tensor, txt_local: torch.Tensor):
    """[Bl,E] x 2 -> [B,E] x 2 on every rank, rank-major: ONE collective on the fused [2, Bl, E] send buffer (received as
    [world, 2, Bl, E]), then one strided copy per modality into the contiguous [B,E] the loss head reads."""
    if not active():
        return img_local, txt_local
    bl, e = img_local.shape
    w = world_size()
    send, recv = gather_send_buffer(bl, e, img_local.device)
    if img_local.data_ptr() != send[0].data_ptr():
        send[0].copy_(img_local)
    if txt_local.data_ptr() != send[1].data_ptr():
        send[1].copy_(txt_local)
    if dist.get_backend() == "gloo" and send.is_cuda:   # rehearsal path only: gloo has no CUDA all_gather_into_tensor
        parts = [torch.empty_like(send) for _ in range(w)]
        dist.all_gather(parts, send)
        recv = torch.stack(parts, dim=0)
    else:
        dist.all_gather_into_tensor(recv.view(w * 2, bl, e), send)     # out = concatenation along dim 0
    return recv[:, 0].reshape(w * bl, e), recv[:, 1].reshape(w * bl, e)   # reshape of a strided view = the one copy


def sharding():
    """(world, rank) of the loss-head sharding: the process group's, or (1, 0)."""
    return (world_size(), get_rank()) if active() else (1, 0)


def exchange_packets(packet: torch.Tensor) -> torch.Tensor:
    """The one collective of the sharded loss head: every rank's statistics packet [P] -> [world, P] (rank-major)."""
    if not active():
        return packet.unsqueeze(0)
    w = world_size()
    if dist.get_backend() == "gloo" and packet.is_cuda:   # rehearsal path only
        parts = [torch.empty_like(packet) for _ in range(w)]
        dist.all_gather(parts, packet)
        return torch.stack(parts, dim=0)
    out = torch.empty(w, packet.numel(), dtype=packet.dtype, device=packet.device)
    dist.all_gather_into_tensor(out.view(-1), packet)
    return out


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def local_rows(full: torch.Tensor):
    """This rank's rows of a gathered [B,E] tensor (the backward of the gather needs no collective)."""
    if not active():
        return full
    a, b = shard_bounds(full.shape[0], get_rank(), world_size())
    return full[a:b].contiguous()


def broadcast_parameters(model):
    if active():
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
        if not active():
            return
        a, b = self.spans[name]
        self.pending.append(dist.all_reduce(self.model.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def wait_all(self):
        for w in self.pending:
            w.wait()
        self.pending = []
