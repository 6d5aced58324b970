#!/usr/bin/env python
"""Launch the bf16 GEMM kernels a few times on the shapes the verdict asks counters for - NT fc1 (+bias+GELU+pre), NT out_proj
(+bias+fp32 residual), NT fc2 plain, TN dW of fc1 - at the ViT-B/32 image tower's local batch 1024 (target of the --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sparsify_clip_amd import ops
dev = "cuda:0"
M = 51200
for (m, n, k), kind in [((M, 3072, 768), "bias+gelu+pre"), ((M, 768, 768), "bias+resid"), ((M, 768, 3072), "plain"), ((M, 3072, 768), "dgelu")]:
    a, b, c, epi, keep = bench.gemm_launch_operands(m, n, k, kind, dev)
    for _ in range(4):
        ops.gemm_bf16_nt(a, b, out=c, epi=epi)
    torch.cuda.synchronize()
    del a, b, c, epi, keep
dy = torch.randn(M, 3072, device=dev).to(torch.bfloat16)
x = torch.randn(M, 768, device=dev).to(torch.bfloat16)
dw = torch.zeros(3072, 768, device=dev)
for _ in range(4):
    ops.gemm_bf16_tn(dy, x, out=dw)
torch.cuda.synchronize()
