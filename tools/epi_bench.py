#!/usr/bin/env python
"""Where does the epilogue time of the c_fc GEMM go?  plain / +second output / +GELU / both / GELU' / resid variants on one shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops
dev = "cuda:0"
def timed(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts)//2]
m, n, k = 51200, 3072, 768
a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
c = torch.empty(m, n, dtype=torch.bfloat16, device=dev); pre = torch.empty_like(c); bias = torch.randn(n, device=dev)
c32 = torch.empty(m, n, device=dev)
variants = {
  "plain bf16 out": (c, None),
  "bias only": (c, ops.make_epilogue(bias=bias)),
  "bias + second output (no act)": (c, ops.make_epilogue(bias=bias, pre_out=pre, ld_aux=n)),
  "bias + GELU (one output)": (c, ops.make_epilogue(bias=bias, act=1)),
  "bias + GELU + second output": (c, ops.make_epilogue(bias=bias, pre_out=pre, act=1, ld_aux=n)),
  "* GELU'(pre) (extra bf16 read)": (c, ops.make_epilogue(dgelu_pre=pre, ld_aux=n)),
  "fp32 out plain": (c32, None),
}
for name, (out, e) in variants.items():
    ms = timed(lambda: ops.gemm_bf16_nt(a, b, out=out, epi=e))
    print(f"{name:40s} {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.1f} TF", flush=True)
# a pure streaming store of the same 315 MB for scale
ms = timed(lambda: c.fill_(0)); print(f"{'fill 315 MB (torch)':40s} {ms*1e3:8.1f} us  {c.numel()*2/ms/1e6:7.1f} GB/s")
ms = timed(lambda: pre.copy_(c)); print(f"{'copy 315 MB (torch)':40s} {ms*1e3:8.1f} us  {2*c.numel()*2/ms/1e6:7.1f} GB/s r+w")
