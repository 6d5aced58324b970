#!/usr/bin/env python
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops
dev = "cuda:0"
m, n, k = 51200, 2304, 768
dy = torch.randn(m, n, device=dev).to(torch.bfloat16); x = torch.randn(m, k, device=dev).to(torch.bfloat16)
dw = torch.zeros(n, k, device=dev)
for _ in range(4):
    ops.gemm_bf16_tn(dy, x, out=dw)
torch.cuda.synchronize()
