#!/usr/bin/env python
"""Bit-reproducibility of the TN weight-gradient GEMM at the training step's shapes, alone and with other GEMMs running
concurrently on other streams (the situation inside a training step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsify_clip_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
shapes = [(51200, 3072, 768), (51200, 768, 3072), (51200, 2304, 768), (51200, 768, 768), (78848, 2048, 512), (78848, 512, 2048), (78848, 1536, 512)]
side = [torch.cuda.Stream(), torch.cuda.Stream()]
xa = torch.randn(51200, 768, device=dev).to(torch.bfloat16); xw = torch.randn(3072, 768, device=dev).to(torch.bfloat16)
xo = torch.empty(51200, 3072, dtype=torch.bfloat16, device=dev)
for r, m, n in shapes:
    a = torch.randn(r, m, device=dev).to(torch.bfloat16); b = torch.randn(r, n, device=dev).to(torch.bfloat16)
    ref_c = torch.zeros(m, n, device=dev); ref_s = torch.zeros(m, device=dev)
    ops.gemm_bf16_tn(a, b, out=ref_c, colsum_out=ref_s)
    torch.cuda.synchronize()
    bad_alone = bad_conc = 0
    for rep in range(8):
        c = torch.zeros(m, n, device=dev); s = torch.zeros(m, device=dev)
        ops.gemm_bf16_tn(a, b, out=c, colsum_out=s)
        torch.cuda.synchronize()
        bad_alone += int(not (torch.equal(c, ref_c) and torch.equal(s, ref_s)))
    for rep in range(8):
        c = torch.zeros(m, n, device=dev); s = torch.zeros(m, device=dev)
        torch.cuda.synchronize()
        for st in side:
            with torch.cuda.stream(st):
                for _ in range(3):
                    ops.gemm_bf16_nt(xa, xw, out=xo)
        ops.gemm_bf16_tn(a, b, out=c, colsum_out=s)
        torch.cuda.synchronize()
        bad_conc += int(not (torch.equal(c, ref_c) and torch.equal(s, ref_s)))
    print(f"TN r={r} [{m}x{n}]: mismatching repeats alone {bad_alone}/8, with concurrent NT GEMMs {bad_conc}/8", flush=True)
    del a, b
