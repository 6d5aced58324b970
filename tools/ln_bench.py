"""Stand-alone timings of the LayerNorm forward / backward kernels at the two towers' shapes (ViT-B/32, local batch 1024):
HBM bytes per launch and the rate they are moved at.  Run on the GPU box:  python tools/ln_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_clip_amd import ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = (("image", 51200, 768), ("text", 78848, 512))
    if os.environ.get("LN_L14"):      # ViT-L/14 at local batch 512 (BASELINE config 5)
        shapes = (("L/14 image", 131584, 1024), ("L/14 text", 39424, 768))
    for name, rows, w in shapes:
        x = torch.randn(rows, w, device=dev)
        g = torch.randn(w, device=dev)
        b = torch.randn(w, device=dev)
        y = torch.empty(rows, w, dtype=torch.bfloat16, device=dev)
        _, mean, rstd = ops.layernorm_fwd(x, g, b, torch.bfloat16, out=y)
        dy = torch.randn(rows, w, device=dev).to(torch.bfloat16)
        dres = torch.randn(rows, w, device=dev)
        dg, db, dc = (torch.zeros(w, device=dev) for _ in range(3))
        n = rows * w
        t_f = timed(lambda: ops.layernorm_fwd(x, g, b, torch.bfloat16, out=y))
        t_b = timed(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres, want_cast=True, dgamma=dg, dbeta=db, accumulate=True, dx_colsum=dc))
        bf, bb = n * 6, n * 16
        print(f"layernorm {name:10s} [{rows}x{w}]: fwd {t_f:7.1f} us ({bf / t_f / 1e6:5.2f} TB/s of {bf / 1e6:.0f} MB)   "
              f"bwd+reduce {t_b:7.1f} us ({bb / t_b / 1e6:5.2f} TB/s of {bb / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
