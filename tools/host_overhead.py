#!/usr/bin/env python
"""Is the step host-bound?  Time until Trainer.step() returns (enqueue only) vs the synchronised step time."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from sparsify_clip_amd.data import synthetic_batch
from sparsify_clip_amd.train import Trainer
dev = torch.device("cuda", 0)
key, cfg = bench.reference_config("experiment_6", "ViT-B-32", 1024, "bf16")
tr = Trainer(cfg, dev, steps_per_epoch=1000); tr.epoch = 1
b = tuple(t.to(dev) for t in synthetic_batch(42, 1024))
for _ in range(2): tr.step(*b)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter(); tr.step(*b); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(json.dumps({"enqueue_ms": [round(x * 1e3, 1) for x in enq], "step_ms": [round(x * 1e3, 1) for x in tot]}))
