#!/usr/bin/env python
"""Per-shape throughput of the bf16 GEMM kernels on the shapes of one ViT-B/32 training step (local batch 1024).
HIP events on the launch stream, median of 5.  SC_GEMM_NT=128 selects the 128x128 2-stage kernel for A/B runs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops

dev = "cuda:0"
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]

rows = {"img": 1024 * 50, "txt": 1024 * 77}
out = []
for tower, (m, w) in {"img": (rows["img"], 768), "txt": (rows["txt"], 512)}.items():
    for name, (n, k) in {"qkv": (3 * w, w), "out": (w, w), "fc1": (4 * w, w), "fc2": (w, 4 * w), "dx_qkv": (w, 3 * w), "dx_fc1": (w, 4 * w)}.items():
        a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        ms = timed(lambda: ops.gemm_bf16_nt(a, b, out=c))
        out.append((f"NT {tower}.{name} [{m}x{n}x{k}] plain", 2.0 * m * n * k / ms / 1e9, ms))
        if name == "fc1":
            bias = torch.randn(n, device=dev); pre = torch.empty_like(c)
            e = ops.make_epilogue(bias=bias, pre_out=pre, act=1, ld_aux=n)
            ms = timed(lambda: ops.gemm_bf16_nt(a, b, out=c, epi=e))
            out.append((f"NT {tower}.{name} +bias+gelu+pre", 2.0 * m * n * k / ms / 1e9, ms))
            e = ops.make_epilogue(dgelu_pre=pre, ld_aux=n)
            ms = timed(lambda: ops.gemm_bf16_nt(a, b, out=c, epi=e))
            out.append((f"NT {tower}.{name} *gelu'(pre)", 2.0 * m * n * k / ms / 1e9, ms))
        if name in ("out", "fc2"):
            bias = torch.randn(n, device=dev); resid = torch.randn(m, n, device=dev); c32 = torch.empty(m, n, device=dev)
            e = ops.make_epilogue(bias=bias, resid=resid, ld_aux=n)
            ms = timed(lambda: ops.gemm_bf16_nt(a, b, out=c32, epi=e))
            out.append((f"NT {tower}.{name} +bias+resid -> fp32", 2.0 * m * n * k / ms / 1e9, ms))
        del a, b, c
    for name, (n, k) in {"dw_qkv": (3 * w, w), "dw_out": (w, w), "dw_fc1": (4 * w, w), "dw_fc2": (w, 4 * w)}.items():
        dy = torch.randn(m, n, device=dev).to(torch.bfloat16); x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        dw = torch.zeros(n, k, device=dev)
        ms = timed(lambda: ops.gemm_bf16_tn(dy, x, out=dw))
        out.append((f"TN {tower}.{name} [{n}x{k}, r={m}]", 2.0 * m * n * k / ms / 1e9, ms))
        del dy, x, dw
for name, tf, ms in out:
    print(f"{name:52s} {tf:8.1f} TFLOP/s  {ms*1e3:8.1f} us", flush=True)
