#!/usr/bin/env python
"""Time of the replicated global-batch loss head (fp32) at the BASELINE batch sizes, per loss stack."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import numpy as np


def philox_embeddings(seed, b, d):
    """Deterministic unit-norm fp32 embedding pair from numpy Philox (timing inputs; the oracle is test infrastructure and is not imported here)."""
    rng = np.random.Generator(np.random.Philox(seed))
    out = []
    for _ in range(2):
        z = rng.standard_normal((b, d), dtype=np.float32)
        out.append(z / np.linalg.norm(z, axis=1, keepdims=True))
    return out


from sparsify_clip_amd.loss_dispatch import step_loss
dev = "cuda:0"
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts)//2]
for exp in ["experiment_2", "experiment_6", "experiment_9"]:
    for b, d in [(1024, 512), (4096, 512), (8192, 512), (4096, 768)]:
        key, cfg = bench.reference_config(exp, "ViT-B-32", b, "bf16")
        i, t = philox_embeddings(1, b, d)
        i, t = torch.tensor(i).to(dev), torch.tensor(t).to(dev)
        ms = timed(lambda: step_loss(cfg, i, t, 0.1, 1, 300, 1000))
        flops = 6.0 * b * b * d + (4.0 * b * b * d if exp != "experiment_2" else 0) * (2 if exp == "experiment_9" else 1)
        print(f"{exp:13s} B={b:5d} D={d}: {ms:7.3f} ms  ({flops/ms/1e9:6.1f} TF/s fp32-MFMA equivalent)", flush=True)
