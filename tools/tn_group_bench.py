#!/usr/bin/env python
"""The four weight gradients of a transformer block as ONE grouped TN launch (sc_gemm_bf16_tn_group) at the step's block shapes:
ViT-B/32 image and text towers at 1024 pairs, ViT-L/14 image tower at 512 pairs.   Run on the GPU box:  python tools/tn_group_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sparsify_clip_amd import ops  # noqa: E402

dev = "cuda:0"
for name, r, w in (("B/32 image", 51200, 768), ("B/32 text", 78848, 512), ("L/14 image", 131584, 1024), ("L/14 text", 39424, 768)):
    shapes = [(w, 4 * w), (4 * w, w), (w, w), (3 * w, w)]
    probs = []
    for (m, n) in shapes:
        probs.append((torch.randn(r, m, device=dev).to(torch.bfloat16), torch.randn(r, n, device=dev).to(torch.bfloat16), torch.zeros(m, n, device=dev)))
    for _ in range(2):
        ops.gemm_bf16_tn_group(probs, beta=1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_bf16_tn_group(probs, beta=1.0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    flops = sum(2.0 * r * m * n for m, n in shapes)
    tiles = sum((m // 256) * (n // 256) for m, n in shapes)
    print(f"TN group {name:10s} [r={r}, width {w}] {tiles:3d} tiles: {us:7.1f} us  ({flops / us / 1e6:6.1f} TFLOP/s, launch + reduce)", flush=True)
    del probs
