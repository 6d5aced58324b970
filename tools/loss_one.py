#!/usr/bin/env python
"""Run the experiment-6 loss head (anchor + lalign + lunif(centroids)) a few times at the global batch 8192 x 512 (target of the
rocprofv3 --pmc passes: FETCH_SIZE / WRITE_SIZE of the pairwise/LSE kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from oracle.loss_head import philox_embeddings
from sparsify_clip_amd.loss_dispatch import step_loss
dev = "cuda:0"
b, d = int(os.environ.get("LOSS_B", "8192")), 512
key, cfg = bench.reference_config("experiment_6", "ViT-B-32", b, "bf16")
i, t = philox_embeddings(1, b, d)
i, t = torch.tensor(i).to(dev), torch.tensor(t).to(dev)
for _ in range(3):
    step_loss(cfg, i, t, 0.1, 1, 300, 1000)
torch.cuda.synchronize()
