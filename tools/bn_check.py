"""Diagnostic: BatchNorm statistics kernels against fp64 on the shapes of the test ResNet."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops
g = torch.Generator().manual_seed(0)
for rows, c in [(6144, 8), (6144, 16), (1536, 16), (1536, 64), (384, 32), (384, 64), (384, 128), (96, 64), (96, 256), (96, 128), (24, 128), (24, 512), (24, 256)]:
    x = (torch.randn(rows, c, generator=g) * torch.rand(c, generator=g) * 3 + torch.randn(c, generator=g) * 4)
    x64 = x.double()
    mean64, var64 = x64.mean(0), x64.var(0, unbiased=False)
    xd = x.cuda()
    mean, rstd = ops.bn_finish(ops.bn_stats(xd), 1, c, rows)
    em = ((mean.cpu().double() - mean64).abs() / (x64.std(0) + 1e-12)).max().item()
    er = ((rstd.cpu().double() * (var64 + 1e-5).sqrt()) - 1).abs().max().item()
    print(f"rows {rows:5d} C {c:4d}: mean err / std {em:.2e}   rstd rel err {er:.2e}")
