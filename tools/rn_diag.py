"""Diagnostic: ModifiedResNet tower gradients (fp32 path) against the oracle evaluated in fp64 and in fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.clip_model import create_model, synthetic_batch
from sparsify_clip_amd.model import ClipModel
DEV = "cuda:0"
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
import sys as _s
NAME, PREC, BATCH = (_s.argv[1:4] + ["test-rn", "fp32", "6"])[:3] if len(_s.argv) > 3 else ("test-rn", "fp32", "6")
BATCH = int(BATCH)
ref = create_model(NAME, seed=3)
sd = ref.state_dict()
gen = torch.Generator().manual_seed(9)
for k in sd:
    if os.environ.get("KEEP_INIT"):
        break
    if k.endswith("bn3.weight") and "layer" in k:
        sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
ref.load_state_dict(sd)
model = ClipModel(NAME, device=DEV, precision=PREC, seed=0)
model.load_state_dict(sd)
images = torch.tensor(synthetic_batch(31, BATCH, ref.cfg)[0])
d_emb = torch.randn(BATCH, ref.cfg["embed_dim"], generator=gen)
out = {}
for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
    r = create_model(NAME, seed=3).to(dt)
    r.load_state_dict(sd)
    r.train()
    w = r.encode_image(images.to(dt))
    w.backward(d_emb.to(dt))
    out[tag] = (w, {k: p.grad for k, p in r.visual.named_parameters()})
rb = create_model(NAME, seed=3); rb.load_state_dict(sd); rb.train()
with torch.autocast("cpu", dtype=torch.bfloat16):
    wb = rb.encode_image(images)
wb.float().backward(d_emb)
out["bf16"] = (wb.float(), {k: p.grad for k, p in rb.visual.named_parameters()})
model.train(); model.zero_grad()
got = model.image_forward(images.to(DEV)); model.image_backward(d_emb.to(DEV)); torch.cuda.synchronize()
print("emb vs f64", rel(got, out["f64"][0]), " torch-f32 vs f64", rel(out["f32"][0], out["f64"][0]), " torch-autocast-bf16 vs f64", rel(out["bf16"][0], out["f64"][0]))
rows = []
for k, g64 in out["f64"][1].items():
    rows.append((rel(model.grad("visual." + k), g64), rel(out["bf16"][1][k], g64), float(g64.norm()), k))
tot_e = sum((model.grad("visual." + k).double().cpu() - g).norm() ** 2 for k, g in out["f64"][1].items()) ** 0.5
tot_n = sum(g.norm() ** 2 for g in out["f64"][1].values()) ** 0.5
print("whole gradient: rel error %.3e" % float(tot_e / tot_n))
for r in (rows if len(_s.argv) > 4 else sorted(rows, reverse=True)[:12]):
    print("hip-vs-f64 %.2e  torch-autocast-bf16-vs-f64 %.2e  |g| %.3e  %s" % r)
