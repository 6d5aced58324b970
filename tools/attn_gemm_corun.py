#!/usr/bin/env python
"""Does a latency-bound kernel hide behind a GEMM when the two share the chip SPATIALLY?  The short-sequence attention backward (persistent,
one workgroup per CU) on G of the 256 CUs on one stream (SC_ATTN_BWD_CUS: a knob that existed for this measurement only - the
script now measures the default grid in every row), the grouped weight-gradient launch of a ViT-B/32 block on another: wall time of
N attention + N GEMM launches one after the other against side by side.   Run on the GPU box:  python tools/attn_gemm_corun.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sparsify_clip_amd import ops  # noqa: E402

dev = "cuda:0"
b, s, w, h = 1024, 50, 768, 12
qkv = torch.randn(b * s, 3 * w, device=dev).to(torch.bfloat16)
d_out = torch.randn(b * s, w, device=dev).to(torch.bfloat16)
r = b * s
shapes = [(w, 4 * w), (4 * w, w), (w, w), (3 * w, w)]
probs = [(torch.randn(r, m, device=dev).to(torch.bfloat16), torch.randn(r, n, device=dev).to(torch.bfloat16), torch.zeros(m, n, device=dev)) for m, n in shapes]
side = torch.cuda.Stream()
N = 12


def attn():
    for _ in range(N):
        ops.attention_bwd(qkv, d_out, b, s, h, False)


def gemm():
    for _ in range(N):
        ops.gemm_bf16_tn_group(probs, beta=1.0)


def timed(f):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gemm()
    attn()
    main.wait_stream(side)


for cus in ("256", "192", "128", "96", "64"):
    os.environ["SC_ATTN_BWD_CUS"] = cus
    attn(); gemm(); both()
    ta, tg, tb = timed(attn), timed(gemm), timed(both)
    print(f"attention bwd on {cus:>3s} CUs: alone {ta:7.1f} us   grouped dW alone {tg:7.1f} us   one after the other {ta + tg:7.1f} us   side by side {tb:7.1f} us", flush=True)
