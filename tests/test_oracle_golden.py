"""CPU: the oracle restatement against the fixtures produced by the reference itself
(oracle/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare the
HIP path with the oracle and with the same fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_json
from oracle import dispatch as D
from oracle import loss_head as L
from oracle import metrics as M
from oracle import schedules as S

RTOL = 2e-6  # fp32 restatement vs fp32 reference: same math, different op order


def _t(a, grad=False):
    return torch.tensor(a, dtype=torch.float32, requires_grad=grad)


def _close(a, b, rtol=RTOL, atol=1e-7):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag", ["rand32x512", "clustered64x768"])
def test_loss_values_and_autograd_grads(golden_small, tag):
    arr, meta = golden_small
    v = meta[tag]["values"]
    img, txt = _t(arr[f"{tag}/img"], True), _t(arr[f"{tag}/txt"], True)

    def check(name, fn, leaves):
        for lf in leaves:
            lf.grad = None
        val = fn()
        val.backward()
        _close(val.item(), v[name])
        for k, lf in enumerate(leaves):
            key = f"{tag}/{name}.g{k}"
            if key in arr:
                _close(lf.grad.numpy(), arr[key], rtol=2e-5, atol=2e-8)

    check("contrastive_T0.1", lambda: L.contrastive_loss(img, txt, 0.1), [img, txt])
    check("contrastive_T0.07", lambda: L.contrastive_loss(img, txt), [img, txt])
    check("lalign", lambda: L.lalign_loss(img, txt), [img, txt])
    check("lalign_alpha1", lambda: L.lalign_loss(img, txt, alpha=1), [img, txt])
    check("lunif_img", lambda: L.lunif_loss(img), [img])
    check("lunif_txt_t3", lambda: L.lunif_loss(txt, t=3), [txt])
    check("lunif_centroids", lambda: L.lunif_centroids(img, txt), [img, txt])
    check("sparsify_img", lambda: L.sparsify_loss(img), [img])
    check("centroid_alignment", lambda: L.centroid_alignment_loss(img, txt), [img, txt])
    temp = torch.nn.Parameter(torch.tensor(0.1))
    val = L.contrastive_loss(img, txt, temperature=temp)
    val.backward()
    _close(temp.grad.item(), v["contrastive_learnableT.g2.value"], rtol=1e-5)
    soft = _t(arr[f"{tag}/soft_targets"])
    _close(L.contrastive_loss_soft(img, txt, soft, 0.1).item(), v["contrastive_roberta_T0.1"])
    norms, cents = L.compute_centroids(_t(arr[f"{tag}/txt"][:5]), _t(arr[f"{tag}/img"][:7]))
    _close(norms.numpy(), arr[f"{tag}/centroid_norms_5x7"])
    _close(cents.sum(-1).numpy(), arr[f"{tag}/centroids_5x7_sum"], atol=1e-6)


@pytest.mark.parametrize("tag", ["rand32x512", "clustered64x768"])
def test_closed_form_fp64_grads_match_reference_autograd(golden_small, tag):
    """The analytic fp64 gradients (what the HIP backward kernels are checked against)."""
    arr, meta = golden_small
    v = meta[tag]["values"]
    img, txt = arr[f"{tag}/img"], arr[f"{tag}/txt"]
    loss, di, dt, dtemp = L.contrastive_grads(img, txt, 0.1)
    _close(loss, v["contrastive_T0.1"])
    _close(di, arr[f"{tag}/contrastive_T0.1.g0"], rtol=2e-5, atol=2e-8)
    _close(dt, arr[f"{tag}/contrastive_T0.1.g1"], rtol=2e-5, atol=2e-8)
    _close(dtemp, v["contrastive_learnableT.g2.value"], rtol=1e-5)
    loss, dx = L.lunif_grads(img, 2)
    _close(loss, v["lunif_img"])
    _close(dx, arr[f"{tag}/lunif_img.g0"], rtol=5e-5, atol=2e-8)
    loss, dx = L.lunif_grads(txt, 3)
    _close(loss, v["lunif_txt_t3"])
    _close(dx, arr[f"{tag}/lunif_txt_t3.g0"], rtol=5e-5, atol=2e-8)
    loss, dx, dy = L.lalign_grads(img, txt, 2)
    _close(loss, v["lalign"])
    _close(dx, arr[f"{tag}/lalign.g0"], rtol=2e-5, atol=1e-9)
    _close(dy, arr[f"{tag}/lalign.g1"], rtol=2e-5, atol=1e-9)
    loss, dx = L.sparsify_grads(img)
    _close(loss, v["sparsify_img"])
    _close(dx, arr[f"{tag}/sparsify_img.g0"], rtol=5e-5, atol=2e-8)


@pytest.mark.parametrize("key", ["b512_d512_rand", "b4096_d512_rand", "b8192_d512_rand", "b4096_d768_rand",
                                 "b2048_d512_clustered"])
def test_large_batches_regenerated_inputs(golden_large, key):
    """Inputs regenerated from the build-owned Philox stream; only the reference's outputs are stored."""
    g = golden_large[key]
    v = g["values"]
    img, txt = L.philox_embeddings(g["seed"], g["b"], g["d"], g["clustered"])
    probe = [(i % g["b"], j % g["d"]) for i, j in g["probe_index"]]
    loss, di, dt, dtemp = L.contrastive_grads(img, txt, 0.1)
    _close(loss, v["contrastive_T0.1"], rtol=1e-5)
    _close(np.linalg.norm(di), v["contrastive_T0.1.g0.norm"], rtol=1e-4)
    _close([di[i, j] for i, j in probe], v["contrastive_T0.1.g0.probe"], rtol=2e-3, atol=1e-9)
    _close(dtemp, v["contrastive_learnableT.g2.value"], rtol=1e-4)
    loss, dx = L.lunif_grads(img, 2)
    _close(loss, v["lunif_img"], rtol=1e-5)
    _close(np.linalg.norm(dx), v["lunif_img.g0.norm"], rtol=1e-4)
    loss, dx, _ = L.lalign_grads(img, txt, 2)
    _close(loss, v["lalign"], rtol=1e-5)
    _close(np.linalg.norm(dx), v["lalign.g0.norm"], rtol=1e-4)
    if g["b"] <= 2048:
        with torch.no_grad():
            _close(L.lunif_loss(_t(img)).item(), v["lunif_img"], rtol=1e-5)
            _close(L.lunif_loss_gram(_t(img)).item(), v["lunif_img"], rtol=1e-5)
            _close(L.lunif_centroids(_t(img), _t(txt)).item(), v["lunif_centroids"], rtol=1e-5)
            _close(L.sparsify_loss(_t(img)).item(), v["sparsify_img"], rtol=1e-5)


def test_schedules_tables():
    sched = load_json("schedules.json")
    for row in sched["beta"]:
        got = [S.get_beta(s, row["total"], row["warmup"], row["ramp"]) for s in row["steps"]]
        assert got == row["values"]
    for row in sched["alpha"]:
        got = [S.get_alpha(s, row["total"], row["warmup"], row["ramp"]) for s in row["steps"]]
        assert got == row["values"]
    for row in sched["lr"]:
        got = [S.lr_multiplier(s, row["warmup_steps"], row["total"], row["only_lunif_epochs"]) for s in row["steps"]]
        assert got == row["values"]
    assert S.get_beta(300, 1000, 20, 50) == sched["known"]["get_beta(300,1000,20,50)"] == 0.8
    assert abs(S.get_alpha(600, 1000, 50, 50) - 1.2) < 1e-12
    assert S.lr_multiplier(0, 280, 1400, 0) == 0.0  # first optimiser step has lr 0


def test_dispatch_against_reference_control_flow(golden_small):
    arr, _ = golden_small
    cfgs = load_json("configs.json")
    disp = load_json("dispatch.json")
    assert len(disp) == 13
    for rel, entry in disp.items():
        cfg = cfgs[rel]
        for row in entry["rows"]:
            img, txt = _t(arr["rand32x512/img"], True), _t(arr["rand32x512/txt"], True)
            loss, beta, alpha = D.compose_loss(cfg, img, txt, cfg["anchor_temperature"], row["epoch"],
                                               row["current_batch"], row["t_total"])
            loss.backward()
            _close(loss.item(), row["loss"], rtol=5e-6)
            _close(img.grad.double().norm().item(), row["dimg_norm"], rtol=2e-5)
            _close(txt.grad.double().norm().item(), row["dtxt_norm"], rtol=2e-5)
            assert (0.0 if beta is None else beta) == row["beta"]
            assert (0.0 if alpha is None else alpha) == row["alpha"]
    # exp-7 and exp-8 share one loss_type string and therefore one formula
    k7 = [k for k in disp if "experiment_7" in k][0]
    k8 = [k for k in disp if "experiment_8" in k][0]
    assert disp[k7]["loss_type"] == disp[k8]["loss_type"]
    assert [r["loss"] for r in disp[k7]["rows"]] == [r["loss"] for r in disp[k8]["rows"]]
    with pytest.raises(KeyError):
        D.lookup("no-such-loss")


def test_metrics_against_reference(golden_metrics):
    arr, v = golden_metrics
    f1, f2 = _t(arr["f1"]), _t(arr["f2"])
    _close(M.numpy_uniformity(f1, f2), v["numpy_uniformity"], rtol=1e-5)
    _close(M.numpy_uniformity(f1, f2), v["sparsify_clip.uniformity"], rtol=1e-5)
    _close(M.torch_uniformity(f1, f2).item(), v["torch_uniformity"], rtol=1e-5)
    _close(M.torch_uniformity1(f1).item(), v["torch_uniformity1"], rtol=1e-5)
    _close(M.torch_uniformity_equivalent(f1).item(), v["torch_uniformity_equivalent"], rtol=1e-5)
    _close(M.uniformity10(f1).item(), v["uniformity10"], rtol=1e-4)
    _close(M.compute_gap(f1, f2), v["compute_gap"], rtol=1e-5)
    _close(M.mean_angular_value(f1), v["mean_angular_value_f1"], rtol=1e-4, atol=1e-8)
    _close(M.mean_cosine_true_pairs(f1, f2), v["mean_distance_of_true_pairs"], rtol=1e-5, atol=1e-8)
    score = f2 @ (0.6 * f2 + 0.4 * f1).t()
    assert M.retrieval_metrics(score, "forward") == v["retrieval_forward"]
    assert M.retrieval_metrics(score, "backward") == v["retrieval_backward"]
    assert score.argmax(dim=1).tolist() == v["top1_forward"]
    assert score.argmax(dim=0).tolist() == v["top1_backward"]
