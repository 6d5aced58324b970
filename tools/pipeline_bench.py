"""Where the input stage spends a batch (run on the GPU box: `python tools/pipeline_bench.py`):
  1. host assembly alone (decode, crop boxes, tokenise, native gather into pinned memory), per phase;
  2. the loader alone (assembly thread + H2D on the copy stream + device crop/resize/normalise), no training step;
  3. H2D of one batch alone."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_clip_amd.data import get_tokenizer                                      # noqa: E402
from sparsify_clip_amd.input_pipeline import DeviceAugLoader, SyntheticCocoDataset    # noqa: E402

B = int(os.environ.get("B", "1024"))
workers = int(os.environ.get("WORKERS", str(min(16, os.cpu_count() or 1))))
dev = torch.device("cuda:0")
ds = SyntheticCocoDataset(12 * B, seed=7)
loader = DeviceAugLoader(ds, B, dev, train=True, seed=7, workers=workers, tokenizer=get_tokenizer("ViT-B-32"))
pinned = torch.empty(loader.cap_bytes, dtype=torch.uint8).pin_memory()
order = np.arange(len(ds))
with ThreadPoolExecutor(workers) as pool:
    loader._assemble(0, order, pool, pinned)
    t = time.perf_counter()
    for b in range(1, 6):
        hb = loader._assemble(b, order, pool, pinned)
    print(f"host assembly alone: {(time.perf_counter() - t) / 5 * 1e3:.1f} ms per batch of {B} ({hb.nbytes / 1e6:.0f} MB of crop pixels, {workers} threads)")
    ids = order[:B]
    t = time.perf_counter(); items = list(pool.map(lambda i: ds[int(i)], ids)); t_dec = time.perf_counter() - t
    rng = loader.batch_rng(0)
    t = time.perf_counter(); samples = [loader._geometry(img, caps, rng) for img, caps in items]; t_geo = time.perf_counter() - t
    t = time.perf_counter(); loader.tokenizer([s[2] for s in samples]).pin_memory(); t_tok = time.perf_counter() - t
    print(f"  fetch {t_dec * 1e3:.1f} ms, crop boxes / flips / caption choice {t_geo * 1e3:.1f} ms, tokenise {t_tok * 1e3:.1f} ms")
gpu = torch.empty(loader.cap_bytes, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    gpu[: hb.nbytes].copy_(pinned[: hb.nbytes], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"H2D alone: {dt * 1e3:.1f} ms per batch = {hb.nbytes / dt / 1e9:.1f} GB/s")
it = iter(loader)
next(it), next(it)
torch.cuda.synchronize()
t = time.perf_counter()
n = 0
for images, caps in it:
    n += 1
torch.cuda.synchronize()
print(f"loader alone (assembly thread + H2D + device augmentation): {(time.perf_counter() - t) / n * 1e3:.1f} ms per batch over {n} batches")

if os.environ.get("TRAIN", "1") == "1":
    # 4. the training step fed by the loader: where each iteration's host time goes, and what the assembly thread achieves meanwhile
    import bench as BENCH
    from sparsify_clip_amd.train import Trainer
    key, cfg = BENCH.reference_config("experiment_6", "ViT-B-32", B, "bf16")
    trainer = Trainer(cfg, dev, steps_per_epoch=1000)
    trainer.epoch = max(1, cfg["only_lunif_epochs"])
    asm_times = []
    inner = loader._assemble

    def timed_assemble(*a, **k):
        t0 = time.perf_counter()
        r = inner(*a, **k)
        asm_times.append(time.perf_counter() - t0)
        return r

    loader._assemble = timed_assemble
    loader.epoch = 0
    it = iter(loader)
    for _ in range(2):
        trainer.step(*next(it))
    torch.cuda.synchronize()
    asm_times.clear()
    t_next, t_step = [], []
    t0 = time.perf_counter()
    for _ in range(8):
        a = time.perf_counter(); batch = next(it); b = time.perf_counter(); trainer.step(*batch); c = time.perf_counter()
        t_next.append(b - a); t_step.append(c - b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"step fed by the loader: {dt * 1e3:.1f} ms per step; host time per iteration: next(batch) {np.mean(t_next) * 1e3:.1f} ms, "
          f"step enqueue {np.mean(t_step) * 1e3:.1f} ms; assembly thread meanwhile {np.mean(asm_times) * 1e3:.1f} ms per batch")
