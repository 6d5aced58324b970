#!/usr/bin/env python
"""EXPERIMENT: do the HBM-bound epilogues of the persistent NT GEMM gain when half of the workgroups run out of phase with the others?
Child processes (the knob is read once per process): SC_GEMM_DESYNC=<cycles> delays the odd workgroups of every XCD before their first
tile.  Shapes: the fp32-residual GEMMs of the step and the GELU forward, with tile tickets."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from sparsify_clip_amd import ops
    dev = "cuda:0"
    def timed(fn, reps=9):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        return sorted(ts)[len(ts) // 2]
    tk = torch.zeros(16, dtype=torch.int32, device=dev)
    res = []
    for (m, n, k), kind in [((51200, 768, 768), "resid"), ((78848, 512, 512), "resid"), ((51200, 768, 3072), "resid"), ((78848, 512, 2048), "resid"),
                            ((51200, 3072, 768), "gelu"), ((51200, 2304, 768), "bias")]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        if kind == "resid":
            resid = torch.randn(m, n, device=dev); c = torch.empty(m, n, device=dev)
            e = ops.make_epilogue(bias=bias, resid=resid, ld_aux=n, tile_tickets=tk)
        elif kind == "gelu":
            c = torch.empty(m, n, dtype=torch.bfloat16, device=dev); pre = torch.empty_like(c)
            e = ops.make_epilogue(bias=bias, pre_out=pre, act=1, ld_aux=n, tile_tickets=tk)
        else:
            c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            e = ops.make_epilogue(bias=bias, tile_tickets=tk)
        ms = timed(lambda: ops.gemm_bf16_nt(a, b, out=c, epi=e))
        res.append(f"{kind} {m}x{n}x{k}: {ms * 1e3:6.1f} us")
        del a, b, c, e
    print(f"desync {os.environ.get('SC_GEMM_DESYNC', '0'):>6s} | " + " | ".join(res), flush=True)
else:
    for d in sys.argv[1:] or ["0", "20000", "40000", "60000", "0"]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SC_GEMM_DESYNC=d), check=True)
