#!/usr/bin/env python
"""ViT-L/14: the long-sequence attention backward (S = 257, 16 heads, batch 512; one workgroup per head, two per CU) beside weight-gradient
GEMMs on another stream: N launches of each alone, one after the other, side by side.   Run on the GPU box:  python tools/attn_gemm_corun_l14.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sparsify_clip_amd import ops  # noqa: E402

dev = "cuda:0"
b, s, w, h = 512, 257, 1024, 16
qkv = torch.randn(b * s, 3 * w, device=dev).to(torch.bfloat16)
d_out = torch.randn(b * s, w, device=dev).to(torch.bfloat16)
lse = torch.empty(b * h * s, device=dev)
out = ops.attention_fwd(qkv, b, s, h, False, lse=lse)
r = b * s
side = torch.cuda.Stream()
N = 8
for name, shapes in (("dW of out_proj + in_proj (64 tiles)", [(w, w), (3 * w, w)]), ("dW of the whole block (192 tiles)", [(w, 4 * w), (4 * w, w), (w, w), (3 * w, w)])):
    probs = [(torch.randn(r, m, device=dev).to(torch.bfloat16), torch.randn(r, n, device=dev).to(torch.bfloat16), torch.zeros(m, n, device=dev)) for m, n in shapes]

    def attn():
        for _ in range(N):
            ops.attention_bwd(qkv, d_out, b, s, h, False, out=out, lse=lse)

    def gemm():
        for _ in range(N):
            ops.gemm_bf16_tn_group(probs, beta=1.0)

    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gemm()
        attn()
        main.wait_stream(side)

    def timed(f):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / N

    attn(); gemm(); both()
    ta, tg, tb = timed(attn), timed(gemm), timed(both)
    print(f"{name}: attention bwd alone {ta:7.1f} us   GEMMs alone {tg:7.1f} us   one after the other {ta + tg:7.1f} us   side by side {tb:7.1f} us", flush=True)
    del probs
