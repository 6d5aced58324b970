"""CPU, gloo, world_size 2: the data-parallel bookkeeping of sparsify_clip_amd.dist (rank-major shard -> fused embedding
all-gather -> replicated global-batch loss -> local gradient rows -> bucketed SUM all-reduce) reproduces the single-process
result.  The arithmetic on each rank is the oracle's (CPU); what is under test is the partition/collective logic, which is
identical under RCCL on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FlatHolder:
    """Minimal stand-in for ClipModel's flat gradient buffer + buckets (GradSync only needs these two attributes)."""

    def __init__(self, params):
        self.params = params
        sizes = [p.numel() for p in params]
        self.flat_grad = torch.zeros(sum(sizes))
        half = sum(sizes[: len(sizes) // 2])
        self.buckets = [("late", (half, sum(sizes))), ("early", (0, half))]   # backward order: later layers first
        self.comm = None

    def pack(self):
        self.flat_grad.copy_(torch.cat([p.grad.reshape(-1) for p in self.params]))


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import loss_head as L
    from oracle.clip_model import create_model, synthetic_batch
    from sparsify_clip_amd import dist as D
    r, lr, w = D.init_process_group("gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.get_rank() == rank
    model = create_model("tiny", seed=4)
    params = [p for n, p in model.named_parameters() if n != "logit_scale"]
    gb = 8
    images_np, tokens_np = synthetic_batch(3, gb, model.cfg)
    images, tokens = torch.tensor(images_np), torch.tensor(tokens_np)
    a, b = D.shard_bounds(gb, rank, world)
    assert (a, b) == (rank * 4, rank * 4 + 4)

    def loss_fn(i, t):
        return L.contrastive_loss(i, t, 0.1) + L.lalign_loss(i, t) + L.lunif_centroids(i, t)

    # --- data-parallel step on this rank's slice
    img_l = L.normalize_rows(model.encode_image(images[a:b]))
    txt_l = L.normalize_rows(model.encode_text(tokens[a:b]))
    img_all, txt_all = D.all_gather_embeddings(img_l.detach(), txt_l.detach())
    assert img_all.shape == (gb, img_l.shape[1])
    assert torch.equal(img_all[a:b], img_l.detach())                      # rank-major order
    img_all.requires_grad_(True), txt_all.requires_grad_(True)
    loss = loss_fn(img_all, txt_all)
    loss.backward()
    torch.autograd.backward([img_l, txt_l], [D.local_rows(img_all.grad), D.local_rows(txt_all.grad)])
    holder = _FlatHolder(params)
    sync = D.GradSync(holder)
    holder.pack()
    sync.bucket_ready("late")
    sync.bucket_ready("early")
    sync.wait_all()
    dp_grad = holder.flat_grad.clone()
    # --- single-process reference on the whole batch
    for p in params:
        p.grad = None
    full = loss_fn(L.normalize_rows(model.encode_image(images)), L.normalize_rows(model.encode_text(tokens)))
    full.backward()
    ref_grad = torch.cat([p.grad.reshape(-1) for p in params])
    losses = [torch.zeros(1) for _ in range(world)]
    tdist.all_gather(losses, loss.detach().reshape(1))
    ok = (abs(loss.item() - full.item()) <= 1e-6 * abs(full.item()) and losses[0].item() == losses[1].item()
          and (dp_grad - ref_grad).norm().item() <= 1e-5 * ref_grad.norm().item())
    # broadcast_parameters: rank 1 starts from different weights and must end with rank 0's
    class M:
        pass
    m = M()
    m.flat = torch.full((10,), float(rank))
    m.refresh_shadows = lambda full: None
    D.broadcast_parameters(m)
    ok = ok and bool((m.flat == 0).all())
    out[rank] = (ok, loss.item(), full.item(), (dp_grad - ref_grad).norm().item() / ref_grad.norm().item())
    tdist.destroy_process_group()


def test_dp2_equals_dp1_on_cpu_gloo():
    world = 2
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        ok, dp_loss, ref_loss, gerr = out[rank]
        assert ok, (rank, dp_loss, ref_loss, gerr)


def test_shard_bounds_and_single_rank_passthrough():
    from sparsify_clip_amd import dist as D
    assert D.shard_bounds(8192, 3, 8) == (3072, 4096)
    with pytest.raises(ValueError):
        D.shard_bounds(10, 0, 4)
    x, y = torch.randn(4, 8), torch.randn(4, 8)
    gx, gy = D.all_gather_embeddings(x, y)
    assert gx is x and gy is y and D.local_rows(x) is x
