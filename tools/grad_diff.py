#!/usr/bin/env python
"""Which gradient slices differ between two identically seeded full-size bf16 training steps?  (bit-reproducibility triage)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_json
from sparsify_clip_amd.config import finalize_config
from sparsify_clip_amd.data import synthetic_batch
from sparsify_clip_amd.train import Trainer
DEV = "cuda:0"
cfgs = load_json("configs.json")
raw = cfgs[[k for k in cfgs if "experiment_6-" in k][0]]
cfg = finalize_config(raw, 0, {"model": "ViT-B-32", "batch_size": 1024, "precision": "bf16"})
images, tokens = [t.to(DEV) for t in synthetic_batch(42, 1024)]
snaps = []
steps = int(os.environ.get("STEPS", "3"))
for rep in range(int(os.environ.get("REPS", "3"))):
    tr = Trainer(cfg, DEV, 1000)
    tr.epoch = 1
    per = []
    for st in range(steps):
        loss = tr.step(images, tokens).item()
        torch.cuda.synchronize()
        per.append((loss, tr.model.flat_grad.clone(), tr.model.flat.clone(), tr.model.visual.bufs["x"][-1].clone(), tr.model.text.bufs["x"][-1].clone()))
    snaps.append(per)
    model = tr.model
def diff_slices(a, b):
    bad = []
    for name in model.slots:
        off, shape = model.slots[name]
        n = 1
        for s_ in shape: n *= s_
        x, y = a[off:off + n], b[off:off + n]
        if not torch.equal(x, y):
            d = (x - y).abs()
            bad.append((name, int((d > 0).sum()), n, float(d.max()), float(x.abs().max())))
    return bad
for rep in range(1, len(snaps)):
    for st in range(steps):
        l0, g0, p0, xi0, xt0 = snaps[0][st]
        l1, g1, p1, xi1, xt1 = snaps[rep][st]
        gb, pb = diff_slices(g0, g1), diff_slices(p0, p1)
        print(f"rep {rep} step {st + 1}: loss {l0!r} vs {l1!r}  img-tower out equal {torch.equal(xi0, xi1)}  txt-tower out equal {torch.equal(xt0, xt1)}  "
              f"grad slices differing {len(gb)}  param slices differing {len(pb)}")
        for x in gb[:12]:
            print("      grad  %-55s %8d / %8d elements, max |diff| %.3e (max |g| %.3e)" % x)
        for x in pb[:6]:
            print("      param %-55s %8d / %8d elements, max |diff| %.3e (max |p| %.3e)" % x)
