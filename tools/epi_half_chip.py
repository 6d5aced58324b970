#!/usr/bin/env python
"""Is the fp32-residual epilogue of the persistent NT GEMM bound by the chip (HBM) or by the CU?  The stamped (diagnostic) instance runs the
out_proj / fc2 shapes once on the whole chip and once with H CUs held by a diagnostic kernel on another stream (tile tickets, so the
remaining workgroups take all tiles).  With fewer CUs in their epilogues at the same time a chip-bound epilogue gets shorter per tile, a
CU-bound one does not.  Per tile: wait (tile start: vmcnt + barrier, i.e. the previous epilogue's stores draining) | main loop | epilogue,
in hundreds of shader cycles.   Run on the GPU box:  python tools/epi_half_chip.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sparsify_clip_amd import ops  # noqa: E402
from sparsify_clip_amd._lib import LIB  # noqa: E402

dev = "cuda:0"
dll = LIB.load()
hook = dll.sc_gemm_bf16_nt_stamps
hook.argtypes, hook.restype = [ctypes.c_void_p], None
occupy = dll.sc_debug_occupy
occupy.argtypes, occupy.restype = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int
side = torch.cuda.Stream()
tickets = torch.zeros(16, dtype=torch.int32, device=dev)
shapes = [((51200, 768, 768), "bias+resid"), ((78848, 512, 512), "bias+resid"), ((51200, 768, 3072), "bias+resid"), ((51200, 3072, 768), "bias+gelu+pre"),
          ((51200, 2304, 768), "bias")]
for (m, n, k), kind in shapes:
    a, b, c, epi, keep = bench.gemm_launch_operands(m, n, k, kind, dev)
    epi.tile_tickets = tickets.data_ptr()
    ntile = 2 * (m // 128) * (n // 256)
    for hog in (0, 64, 128, 192):
        st = torch.zeros(ntile, 4, dtype=torch.int64, device=dev)
        ops.gemm_bf16_nt(a, b, out=c, epi=epi)
        torch.cuda.synchronize()
        if hog:
            with torch.cuda.stream(side):
                occupy(hog, 8000, ctypes.c_void_p(side.cuda_stream))
            torch.cuda._sleep(400000)
        hook(ctypes.c_void_p(st.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_bf16_nt(a, b, out=c, epi=epi)
        e1.record()
        torch.cuda.synchronize()
        hook(None)
        s = st.cpu().double()
        s = s[s[:, 3] > 0]
        wait, loop, ep = (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0, (s[:, 3] - s[:, 2]) / 100.0
        print(f"{kind:14s} [{m}x{n}x{k}] {hog:3d} CUs held: {len(s):5d} tiles  wait {wait.median():7.1f}  loop {loop.median():7.1f}  epilogue {ep.median():7.1f}"
              f"   (sum {(wait + loop + ep).median():7.1f}; launch {e0.elapsed_time(e1) * 1e3:7.1f} us)", flush=True)
        assert int(tickets.abs().sum()) == 0
    del a, b, c, epi, keep
