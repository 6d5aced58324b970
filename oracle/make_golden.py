#!/usr/bin/env python
"""Generate tests/golden/* by running the REFERENCE's own functions (build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference (/root/reference, read-only) is imported with inert stand-ins registered in
sys.modules for the eight off-path third-party modules that are absent here (torchvision,
openTSNE, wandb, open_clip, umap, sentence_transformers); its module top level is imports,
defs and a warnings filter only.  Nothing of the reference is copied: the fixtures hold
inputs (or the Philox seed that regenerates them) and the reference's outputs.

The loss_type dispatch is inline in the reference's train_model (sparsify_clip.py:778-938) and
cannot be called; its if/elif statement is pulled out of the parsed AST at run time and executed
against the reference's own loss functions, so the dispatch fixture is the reference's control
flow, not ours.
"""
from __future__ import annotations

import ast
import glob
import io
import json
import os
import sys
from contextlib import redirect_stdout
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.loss_head import philox_embeddings  # noqa: E402  (input generator only)

for _m in ["torchvision", "torchvision.datasets", "torchvision.transforms", "openTSNE", "wandb",
           "open_clip", "umap", "sentence_transformers"]:
    sys.modules[_m] = MagicMock()
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import sparsify_clip as ref  # noqa: E402
import uniformity as ref_unif  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PROBE = [(0, 0), (0, 7), (1, 3), (5, 100), (17, 255), (31, 511), (13, 64), (29, 300)]


def _t(a, grad=True):
    return torch.tensor(a, dtype=torch.float32, requires_grad=grad)


def loss_bundle(img_np, txt_np, full_grads):
    """All loss-head values (+ gradients) the reference produces for one embedding pair."""
    out = {}

    def run(name, fn, *leaves):
        for lf in leaves:
            lf.grad = None
        val = fn()
        val.backward()
        out[name] = float(val.item())
        for tag, lf in zip(["g0", "g1", "g2"], leaves):
            g = lf.grad.detach().numpy()
            if full_grads and g.ndim:
                out[f"{name}.{tag}"] = g.copy()
            out[f"{name}.{tag}.norm"] = float(np.linalg.norm(g.astype(np.float64)))
            if g.ndim == 2:
                out[f"{name}.{tag}.probe"] = [float(g[i % g.shape[0], j % g.shape[1]]) for i, j in PROBE]
            else:
                out[f"{name}.{tag}.value"] = float(g)

    img, txt = _t(img_np), _t(txt_np)
    run("contrastive_T0.1", lambda: ref.contrastive_loss(img, txt, 0.1), img, txt)
    run("contrastive_T0.07", lambda: ref.contrastive_loss(img, txt), img, txt)
    temp = torch.nn.Parameter(torch.tensor(0.1, dtype=torch.float32))
    run("contrastive_learnableT", lambda: ref.contrastive_loss(img, txt, temperature=temp), img, txt, temp)
    run("lalign", lambda: ref.lalign_loss(img, txt), img, txt)
    run("lalign_alpha1", lambda: ref.lalign_loss(img, txt, alpha=1), img, txt)
    run("lunif_img", lambda: ref.lunif_loss(img), img)
    run("lunif_txt_t3", lambda: ref.lunif_loss(txt, t=3), txt)
    run("lunif_centroids",
        lambda: ref.lunif_loss(F.normalize(ref.compute_centroids_only(img, txt), dim=-1)), img, txt)
    run("sparsify_img", lambda: ref.sparsify_loss(img), img)
    run("centroid_alignment", lambda: ref.centroid_alignment_loss(img, txt), img, txt)
    return out


def dump_bundle(bundle):
    arrays = {k: v for k, v in bundle.items() if isinstance(v, np.ndarray)}
    scalars = {k: v for k, v in bundle.items() if not isinstance(v, np.ndarray)}
    return arrays, scalars


def make_loss_fixtures():
    # (1) small, explicit inputs + full gradients
    arrays, meta = {}, {}
    for tag, (seed, b, d, clustered) in {"rand32x512": (1234, 32, 512, False),
                                         "clustered64x768": (77, 64, 768, True)}.items():
        img, txt = philox_embeddings(seed, b, d, clustered)
        arr, sc = dump_bundle(loss_bundle(img, txt, full_grads=True))
        arrays[f"{tag}/img"], arrays[f"{tag}/txt"] = img, txt
        for k, v in arr.items():
            arrays[f"{tag}/{k}"] = v
        meta[tag] = {"seed": seed, "b": b, "d": d, "clustered": clustered, "values": sc}
        # soft-target (roberta) variant and the unused all-pairs centroid helper
        soft = torch.softmax(_t(img, False) @ _t(txt, False).t() * 5, dim=1)
        i2, t2 = _t(img), _t(txt)
        v = ref.contrastive_loss_roberta(i2, t2, soft, 0.1)
        v.backward()
        meta[tag]["values"]["contrastive_roberta_T0.1"] = float(v.item())
        arrays[f"{tag}/contrastive_roberta_T0.1.g0"] = i2.grad.numpy().copy()
        arrays[f"{tag}/soft_targets"] = soft.numpy()
        norms, cents = ref.compute_centroids(_t(txt[:5], False), _t(img[:7], False))
        arrays[f"{tag}/centroid_norms_5x7"] = norms.numpy()
        arrays[f"{tag}/centroids_5x7_sum"] = cents.sum(dim=-1).numpy()
    np.savez_compressed(os.path.join(OUT, "loss_small.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "loss_small.json"), "w"), indent=1)

    # (2) large: inputs regenerated from the Philox seed, outputs only
    large = {}
    for seed, b, d, clustered in [(1234, 512, 512, False), (1234, 4096, 512, False), (1234, 8192, 512, False),
                                 (1234, 4096, 768, False), (99, 2048, 512, True)]:
        img, txt = philox_embeddings(seed, b, d, clustered)
        _, sc = dump_bundle(loss_bundle(img, txt, full_grads=False))
        large[f"b{b}_d{d}_{'clustered' if clustered else 'rand'}"] = {
            "seed": seed, "b": b, "d": d, "clustered": clustered, "probe_index": PROBE, "values": sc}
        print("large", b, d, clustered, {k: v for k, v in sc.items() if "." not in k or k.endswith("0.1")})
    json.dump(large, open(os.path.join(OUT, "loss_large.json"), "w"), indent=1)


def make_schedule_fixtures():
    sched = {"beta": [], "alpha": [], "lr": []}
    for total in (1400, 46200):
        for (w, r) in ((20, 50), (50, 50)):
            steps = sorted(set(list(range(0, total + 1, max(1, total // 97))) + [1, 2, total - 1, total]))
            sched["beta"].append({"total": total, "warmup": w, "ramp": r, "steps": steps,
                                  "values": [ref.get_beta(s, total, w, r) for s in steps]})
            sched["alpha"].append({"total": total, "warmup": w, "ramp": r, "steps": steps,
                                   "values": [ref.get_alpha(s, total, w, r) for s in steps]})
        for only_lunif in (0, 1):
            opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
            warm = int(0.2 * total)
            sch = ref.get_cosine_schedule_with_warmup(opt, warm, total, config={"only_lunif_epochs": only_lunif})
            lam = sch.lr_lambdas[0]
            steps = sorted(set(list(range(0, total + 1, max(1, total // 97))) + [0, 1, 461, 462, 463, warm - 1, warm, warm + 1]))
            sched["lr"].append({"total": total, "warmup_steps": warm, "only_lunif_epochs": only_lunif,
                                "steps": steps, "values": [lam(s) for s in steps]})
    sched["known"] = {"get_beta(300,1000,20,50)": ref.get_beta(300, 1000, 20, 50),
                      "get_alpha(600,1000,50,50)": ref.get_alpha(600, 1000, 50, 50)}
    json.dump(sched, open(os.path.join(OUT, "schedules.json"), "w"))


def yaml_files():
    return sorted(glob.glob(os.path.join(REF, "experiments_configs", "*.yaml")) +
                  glob.glob(os.path.join(REF, "ablatation_configs", "*.yaml")))


def make_config_fixtures():
    parsed = {}
    for f in yaml_files():
        cfg = yaml.safe_load(open(f))
        rel = os.path.relpath(f, REF)
        if cfg is None:
            parsed[rel] = None
            continue
        cfg["learning_rate"] = float(cfg["learning_rate"])  # reference sparsify_clip.py:1141
        parsed[rel] = cfg
    json.dump(parsed, open(os.path.join(OUT, "configs.json"), "w"), indent=1)
    return parsed


def reference_dispatch_stmt():
    """The `if config["loss_type"] == ...` statement of train_model, from the reference's AST."""
    tree = ast.parse(open(os.path.join(REF, "sparsify_clip.py")).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "train_model")
    for node in ast.walk(fn):
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare):
            src = ast.unparse(node.test)
            if src.startswith("config['loss_type'] == 'anchor'") and "roberta" not in src:
                return compile(ast.Module(body=[node], type_ignores=[]), "<reference dispatch>", "exec")
    raise RuntimeError("dispatch statement not found")


def make_dispatch_fixtures(parsed):
    code = reference_dispatch_stmt()
    img_np, txt_np = philox_embeddings(1234, 32, 512, False)
    out = {}
    for rel, cfg in parsed.items():
        if cfg is None:
            continue
        rows = []
        for epoch, current_batch, t_total in [(0, 1, 1000), (0, 300, 1000), (1, 300, 1000), (3, 650, 1000), (9, 990, 1000)]:
            img, txt = _t(img_np), _t(txt_np)
            ns = {"config": cfg, "image_embeds": img, "text_embeds": txt, "temperature": cfg["anchor_temperature"],
                  "epoch": epoch, "current_batch": current_batch, "t_total": t_total, "F": F, "beta": 0.0, "alpha": 0.0,
                  "loss": 0, **{k: getattr(ref, k) for k in ["contrastive_loss", "lunif_loss", "lalign_loss",
                                                             "compute_centroids_only", "get_beta", "get_alpha"]}}
            exec(code, ns)
            ns["loss"].backward()
            rows.append({"epoch": epoch, "current_batch": current_batch, "t_total": t_total,
                         "loss": float(ns["loss"].item()), "beta": float(ns["beta"]), "alpha": float(ns["alpha"]),
                         "dimg_norm": float(img.grad.double().norm()), "dtxt_norm": float(txt.grad.double().norm())})
        out[rel] = {"loss_type": cfg["loss_type"], "rows": rows}
    json.dump(out, open(os.path.join(OUT, "dispatch.json"), "w"), indent=1)


def make_metric_fixtures():
    g = torch.Generator().manual_seed(1234)
    f1 = F.normalize(torch.randn(256, 512, generator=g), dim=-1)
    f2 = F.normalize(torch.randn(256, 512, generator=g), dim=-1)
    np.savez_compressed(os.path.join(OUT, "metric_inputs.npz"), f1=f1.numpy(), f2=f2.numpy())
    with redirect_stdout(io.StringIO()):  # numpy_uniformity prints covariance.shape (uniformity.py:108)
        nu = ref_unif.numpy_uniformity(f1, f2)
    vals = {"numpy_uniformity": float(nu), "torch_uniformity": float(ref_unif.torch_uniformity(f1, f2)),
            "torch_uniformity1": float(ref_unif.torch_uniformity1(f1)),
            "torch_uniformity_equivalent": float(ref_unif.torch_uniformity_equivalent(f1)),
            "uniformity10": float(ref_unif.uniformity10(f1)),
            "sparsify_clip.uniformity": float(ref.uniformity(f1, f2)),
            "compute_gap": ref.compute_gap(f1, f2),
            "mean_angular_value_f1": ref.compute_mean_angular_value_of_a_modality(f1),
            "mean_distance_of_true_pairs": ref.mean_distance_of_true_pairs(f1, f2)}
    # retrieval on a matrix with a planted, noisy diagonal
    score = f2 @ (0.6 * f2 + 0.4 * f1).t()
    ids = list(range(256))
    vals["retrieval_forward"] = ref.compute_metric_ret(score, ids, ids, "forward")
    vals["retrieval_backward"] = ref.compute_metric_ret(score, ids, ids, "backward")
    vals["top1_forward"] = score.argmax(dim=1).tolist()
    vals["top1_backward"] = score.argmax(dim=0).tolist()
    json.dump(vals, open(os.path.join(OUT, "metrics.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    parsed = make_config_fixtures()
    make_schedule_fixtures()
    make_dispatch_fixtures(parsed)
    make_metric_fixtures()
    make_loss_fixtures()
    print("fixtures written to", OUT)
