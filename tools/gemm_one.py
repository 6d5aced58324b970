#!/usr/bin/env python
"""Launch the bf16 NT GEMM on two representative shapes of the ViT-B/32 step a few times (target of the rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops
dev = "cuda:0"
for (m, n, k) in [(51200, 768, 3072), (51200, 3072, 768)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    for _ in range(4):
        ops.gemm_bf16_nt(a, b, out=c)
    torch.cuda.synchronize()
