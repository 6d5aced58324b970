#!/usr/bin/env python
"""Launch ONCE every bf16 NT GEMM of one ViT-B/32 training step at local batch 1024, with the epilogue the step fuses into it
(the launch set bench.py's roofline replays).  Target of the rocprofv3 --pmc passes that measure HBM traffic per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from sparsify_clip_amd import ops
from sparsify_clip_amd.model import CONFIGS
cfg = CONFIGS["ViT-B-32"]
dev = "cuda:0"
for (m, n, k), kind, reps in bench.step_gemm_launches(cfg, 1024, 3 * cfg["patch"] ** 2, cfg["image_size"] // cfg["patch"]):
    a, b, c, epi, keep = bench.gemm_launch_operands(m, n, k, kind, dev)
    torch.cuda.synchronize()
    for _ in range(reps):
        ops.gemm_bf16_nt(a, b, out=c, epi=epi)
    torch.cuda.synchronize()
    del a, b, c, epi, keep
