#!/usr/bin/env python
"""Run the experiment-6 loss head (anchor + lalign + lunif(centroids)) a few times at the global batch 8192 x 512 (target of the
rocprofv3 --pmc passes: FETCH_SIZE / WRITE_SIZE of the pairwise/LSE kernels)."""
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
b, d = int(os.environ.get("LOSS_B", "8192")), 512
key, cfg = bench.reference_config("experiment_6", "ViT-B-32", b, "bf16")
i, t = philox_embeddings(1, b, d)
i, t = torch.tensor(i).to(dev), torch.tensor(t).to(dev)
for _ in range(3):
    step_loss(cfg, i, t, 0.1, 1, 300, 1000)
torch.cuda.synchronize()
