#!/usr/bin/env python
"""From the rocprofv3 kernel trace of tools/gemm_replay.py (one stream, every bf16 NT GEMM call of one ViT-B/32 step at local batch
1024 with its fused epilogue): kernel time per logical sc_gemm_bf16_nt call and the TFLOP/s it implies - the figure that must agree
with bench.py's roofline (`avg_launch_us`, `achieved`)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = sys.argv[1]
if not os.path.exists(path):
    c = glob.glob(os.path.join(os.path.dirname(path), "**", os.path.basename(path)), recursive=True)
    path = c[0]
import bench
from sparsify_clip_amd.model import CONFIGS
cfg = CONFIGS["ViT-B-32"]
launches = bench.step_gemm_launches(cfg, 1024, 3 * cfg["patch"] ** 2, cfg["image_size"] // cfg["patch"])
calls = sum(r for _, _, r in launches)
flops = sum(2.0 * m * n * k * r for (m, n, k), _, r in launches)
rows = [r for r in csv.DictReader(open(path)) if "gemm_bf16_nt" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
tot = sum(dur)
print(f"NT GEMM kernels in the trace: {len(rows)} dispatches for {calls} logical calls; total {tot / 1e3:.2f} ms")
print(f"average per call {tot / calls:.1f} us  ->  {flops / (tot * 1e-6) / 1e12:.1f} TFLOP/s = {flops / (tot * 1e-6) / 1e12 / 2500:.4f} of the 2.5 PF bf16 MFMA peak")
# per launch kind, in issue order: the replay runs each (shape, epilogue) `reps` times back to back (after one warm-up in operands())
names = {}
for r, d in zip(rows, dur):
    key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44], r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    names.setdefault(key, []).append(d)
for (k, g), v in names.items():
    print(f"  {k:44s} grid {g:>8s}: {len(v):4d} dispatches, avg {sum(v) / len(v):8.1f} us")
