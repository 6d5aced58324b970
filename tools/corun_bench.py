"""The persistent NT GEMM with part of the GPU taken by another stream's kernel (a stand-in for an RCCL collective during the
backward pass): a diagnostic kernel keeps H CUs busy while ten fc1-shaped GEMMs run; fixed tile lists against tile tickets.
Run on the GPU box:  python tools/corun_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsify_clip_amd import ops  # noqa: E402
from sparsify_clip_amd._lib import LIB  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    dll = LIB.load()
    occupy = dll.sc_debug_occupy
    occupy.argtypes, occupy.restype = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int
    torch.manual_seed(0)
    side = torch.cuda.Stream()
    tickets = torch.zeros(16, dtype=torch.int32, device=dev)
    for (m, n, k) in ((51200, 3072, 768), (51200, 768, 3072), (78848, 2048, 512)):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = torch.randn(n, k, device=dev).to(torch.bfloat16)
        c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        ref = None
        for hog in (0, 16, 32, 64):
            row = []
            for dyn in (False, True):
                epi = ops.make_epilogue(tile_tickets=tickets if dyn else None)
                for _ in range(2):
                    ops.gemm_bf16_nt(a, b, out=c, epi=epi)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if hog:
                    with torch.cuda.stream(side):
                        occupy(hog, 6000, ctypes.c_void_p(side.cuda_stream))
                    torch.cuda._sleep(200000)   # let the occupying workgroups settle before the GEMMs are enqueued
                e0.record()
                for _ in range(10):
                    ops.gemm_bf16_nt(a, b, out=c, epi=epi)
                e1.record()
                torch.cuda.synchronize()
                row.append(e0.elapsed_time(e1) * 100.0)   # us per GEMM
                if ref is None:
                    ref = c.clone()
                assert torch.equal(c, ref), "result depends on the tile order"
            assert int(tickets.abs().sum()) == 0, "tickets not left zero"
            flops = 2.0 * m * n * k
            print(f"NT [{m}x{n}x{k}] {hog:3d} CUs taken: fixed lists {row[0]:7.1f} us ({flops / row[0] / 1e6:6.1f} TF/s)   "
                  f"tickets {row[1]:7.1f} us ({flops / row[1] / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
