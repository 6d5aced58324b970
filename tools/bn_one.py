#!/usr/bin/env python
"""One BatchNorm layer forward + backward and one implicit 3x3 convolution (forward, input gradient, weight gradient) at RN50's first-stage
shape (batch 256, 56 x 56, 64 channels; BN_C / BN_HW to change) a few times - the target of the rocprofv3 --pmc passes of `tools/gpu.sh pmc_bn`
(FETCH_SIZE / WRITE_SIZE against the algorithmic bytes printed at the end) and of plain timing."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                     # noqa: E402

from sparsify_clip_amd import ops                # noqa: E402

dev = "cuda:0"
b, hw, c = 256, int(os.environ.get("BN_HW", "56")), int(os.environ.get("BN_C", "64"))
rows = b * hw * hw
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(rows, c, device=dev, generator=g).to(torch.bfloat16)
dy = torch.randn(rows, c, device=dev, generator=g).to(torch.bfloat16)
gamma, beta = torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g)
w = (torch.randn(c, c, 3, 3, device=dev, generator=g) * 0.05)
w_taps = w.permute(0, 2, 3, 1).reshape(c, 9 * c).to(torch.bfloat16).contiguous()
w_dx = w.flip(2, 3).permute(1, 2, 3, 0).reshape(c, 9 * c).to(torch.bfloat16).contiguous()
dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
flat, img = ops.halo_buffer(b, hw, hw, c, torch.bfloat16, dev)
flatd, imgd = ops.halo_buffer(b, hw, hw, c, torch.bfloat16, dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
for it in range(3):
    ev[0].record()
    mean, rstd = ops.bn_finish(ops.bn_stats(x), 1, c, rows)
    ev[1].record()
    ops.bn_apply(x, mean, rstd, gamma, beta, True, halo=(img, hw, hw))
    ev[2].record()
    z = ops.conv3x3_bf16(img, w_taps, b, hw, hw)
    ev[3].record()
    sums = ops.bn_bwd_stats(dy, None, x, mean, rstd, True, gamma, beta)
    ev[4].record()
    ops.bn_bwd_apply(dy, None, x, mean, rstd, gamma, sums, rows, True, dg, db, False, beta=beta, halo=(imgd, hw, hw))
    ev[5].record()
    dxc = ops.conv3x3_bf16(imgd, w_dx, b, hw, hw)
    ev[6].record()
    dw = ops.conv3x3_dw_bf16(imgd, flat, b, hw, hw)
    ev[7].record()
torch.cuda.synchronize()
a = rows * c * 2 / 1e6
halo = b * (hw + 2) * (hw + 2) * c * 2 / 1e6
names = ["statistics (read A)", "apply -> bordered image (read A, write A')", "implicit conv forward (read A', write A)", "backward sums (read 2A)",
         "backward apply -> bordered image (read 2A, write A')", "implicit conv input gradient (read A', write A)", "implicit conv weight gradient (read 2A')"]
alg = [a, a + halo, halo + a, 2 * a, 2 * a + halo, halo + a, 2 * halo]
print(f"rows {rows}, C {c}: activation A = {a:.1f} MB, bordered A' = {halo:.1f} MB")
for k, (n, by) in enumerate(zip(names, alg)):
    us = ev[k].elapsed_time(ev[k + 1]) * 1e3
    print(f"  {n:58s} {us:7.1f} us   algorithmic {by:7.1f} MB = {by / us:.2f} TB/s")
