#!/usr/bin/env python
"""Attention forward / backward at the step's two shapes (image: S=50 x 12 heads, text: S=77 x 8 heads, causal), HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops
dev = "cuda:0"
def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
shapes = {"image": (1024, 50, 768, 12, False), "text": (1024, 77, 512, 8, True)}
if os.environ.get("ATTN_L14"):
    shapes = {"vit-l/14 image": (512, 257, 1024, 16, False)}
for name, (b, s, w, h, causal) in shapes.items():
    qkv = torch.randn(b * s, 3 * w, device=dev).to(torch.bfloat16)
    do = torch.randn(b * s, w, device=dev).to(torch.bfloat16)
    out = ops.attention_fwd(qkv, b, s, h, causal)
    f = timed(lambda: ops.attention_fwd(qkv, b, s, h, causal))
    g = timed(lambda: ops.attention_bwd(qkv, do, b, s, h, causal))
    extra = ""
    if ops.attention_uses_stats(qkv.dtype, s):      # the flash-attention form: statistics from the forward, one sweep in the backward's pass 1
        lse = torch.empty(b * h * s, device=dev)
        out = ops.attention_fwd(qkv, b, s, h, causal, lse=lse)
        f2 = timed(lambda: ops.attention_fwd(qkv, b, s, h, causal, lse=lse))
        g2 = timed(lambda: ops.attention_bwd(qkv, do, b, s, h, causal, out=out, lse=lse))
        extra = f"   | with statistics: fwd {f2*1e3:7.1f} us  bwd {g2*1e3:7.1f} us"
    fb = (qkv.numel() + out.numel()) * 2; bb = (2 * qkv.numel() + do.numel()) * 2
    print(f"attention {name:5s} S={s}: fwd {f*1e3:7.1f} us ({fb/f/1e9:6.2f} TB/s of {fb/1e6:.0f} MB)   bwd {g*1e3:7.1f} us ({bb/g/1e9:6.2f} TB/s of {bb/1e6:.0f} MB){extra}", flush=True)
