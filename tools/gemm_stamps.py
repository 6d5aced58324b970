#!/usr/bin/env python
"""Where a tile of the persistent NT GEMM spends its time: runs the DIAGNOSTIC (stamped) instances of gemm_bf16_nt_pers_kernel on the
step's shapes and prints, per shape, the medians over tiles of  wait (tile start: vmcnt + barrier) | main loop | epilogue  in
microseconds (s_memtime runs at 100 MHz).  Shares, not absolute run time: the stamps add fences (cdna_hip_programming.md 7)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sparsify_clip_amd import ops
from sparsify_clip_amd._lib import LIB
dev = "cuda:0"
dll = LIB.load()
hook = dll.sc_gemm_bf16_nt_stamps
hook.argtypes, hook.restype = [ctypes.c_void_p], None
shapes = [((51200, 3072, 768), "bias+gelu+pre"), ((51200, 768, 768), "bias+resid"), ((51200, 2304, 768), "bias"), ((51200, 768, 3072), "plain"),
          ((51200, 768, 3072), "bias+resid"), ((51200, 3072, 768), "dgelu"), ((78848, 512, 512), "bias+resid"), ((78848, 2048, 512), "bias+gelu+pre")]
for (m, n, k), kind in shapes:
    a, b, c, epi, keep = bench.gemm_launch_operands(m, n, k, kind, dev)
    tiles = 2 * (m // 128) * (n // 256)          # upper bound on whole + half tiles
    st = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
    ops.gemm_bf16_nt(a, b, out=c, epi=epi)          # warm-up, production instance
    torch.cuda.synchronize()
    hook(ctypes.c_void_p(st.data_ptr()))
    ops.gemm_bf16_nt(a, b, out=c, epi=epi)
    torch.cuda.synchronize()
    hook(None)
    s = st.cpu().double()
    s = s[s[:, 3] > 0]
    wait, loop, epi_t = (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0, (s[:, 3] - s[:, 2]) / 100.0
    span = (s[:, 3].max() - s[:, 0].min()) / 100.0
    ideal = 2.0 * 256 * 256 * k / (2.5e15 / 256) * 1e6
    print(f"{kind:14s} [{m}x{n}x{k}] {len(s):5d} stamped tiles: wait {wait.median():6.2f}  loop {loop.median():6.2f}  epilogue {epi_t.median():6.2f} us "
          f"(per tile; MFMA time at peak {ideal:5.2f} us), kernel span {span:7.1f} us; loop p10/p90 {loop.quantile(0.1):.2f}/{loop.quantile(0.9):.2f}, "
          f"epilogue p10/p90 {epi_t.quantile(0.1):.2f}/{epi_t.quantile(0.9):.2f}", flush=True)
    # per workgroup (static tile list: virtual block id % grid): when does it finish, how much of its span is spent in tiles
    idx = torch.nonzero(st[:, 3] > 0).squeeze(1).cpu()
    grid = min(256, len(s))
    t_begin = s[:, 0].min()
    end = torch.zeros(grid, dtype=torch.float64)
    busy = torch.zeros(grid, dtype=torch.float64)
    for row, vb in zip(s, idx.tolist()):
        w = vb % grid
        end[w] = max(end[w], (row[3] - t_begin) / 100.0)
        busy[w] += (row[3] - row[0]) / 100.0
    print(f"               workgroup end times (x100 cycles after the first stamp): min {end.min():.0f}  median {end.median():.0f}  max {end.max():.0f};  "
          f"time inside tiles per workgroup: median {busy.median():.0f}  max {busy.max():.0f}", flush=True)
    del a, b, c, epi, keep, st
