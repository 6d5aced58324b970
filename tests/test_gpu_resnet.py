"""GPU parity tests of the ModifiedResNet ("RN50") image tower: the conv.hip kernels against torch (fp64 on the CPU), and the tower
(forward, backward, BatchNorm running statistics, training steps) against the oracle's torch.nn restatement on identical weights.
open_clip itself is absent offline: the oracle is "parity unpinned" for the encoders (oracle/clip_model.py), its RN50 parameter
count (102 007 137) matches the published model."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sparsify_clip_amd import ops as o
    return o


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).norm() / want.norm().clamp_min(1e-30)).item()


def nhwc_rows(x):      # [B,C,H,W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("stride,c,h", [(1, 8, 6), (2, 8, 8), (2, 3, 10), (1, 16, 5)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_lowering(ops, stride, c, h, dtype):
    """im2col3x3 + GEMM == F.conv2d(padding=1, stride); col2im3x3 of (dY W) == the input gradient; NCHW image input for the stem."""
    g = torch.Generator().manual_seed(1)
    b, cout, w = 3, 8, h + 2
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(cout, c, 3, 3, generator=g) * 0.2
    if dtype == torch.bfloat16:
        x, wt = x.to(dtype).float(), wt.to(dtype).float()
    x64, w64 = x.double().requires_grad_(True), wt.double()
    y = F.conv2d(x64, w64, stride=stride, padding=1)
    ho, wo = y.shape[2], y.shape[3]
    kpad = (9 * c + 63) // 64 * 64 if dtype == torch.bfloat16 else 9 * c
    cols = ops.im2col3x3(nhwc_rows(x).to(dtype).to(DEV), b, h, w, c, stride, kpad, dtype)
    assert cols.shape == (b * ho * wo, kpad)
    wg = torch.zeros(cout, kpad)
    wg[:, : 9 * c] = wt.permute(0, 2, 3, 1).reshape(cout, 9 * c)
    got = cols.float().cpu().double() @ wg.double().t()
    assert rel(got, nhwc_rows(y)) < 1e-6
    if c == 3:      # the stem reads the fp32 image tensor directly
        cols2 = ops.im2col3x3(x.to(DEV), b, h, w, c, stride, kpad, dtype, nchw_images=True)
        assert torch.equal(cols2, cols)
    dy = torch.randn(b, cout, ho, wo, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.to(dtype).float()
    y.backward(dy.double())
    dcols = (nhwc_rows(dy).double() @ wg.double()).to(dtype).to(DEV)          # [rows_out, kpad]
    dx = ops.col2im3x3(dcols, b, h, w, c, stride, kpad)
    assert rel(dx, nhwc_rows(x64.grad)) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_avgpool_and_attnpool_tokens(ops, dtype):
    g = torch.Generator().manual_seed(2)
    b, c, h, w = 2, 8, 6, 4
    x = torch.randn(b, c, h, w, generator=g).to(dtype).float()
    x64 = x.double().requires_grad_(True)
    y = F.avg_pool2d(x64, 2)
    got = ops.avgpool_fwd(nhwc_rows(x).to(dtype).to(DEV), b, h, w, c, 2)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert rel(got, nhwc_rows(y)) < tol
    dy = torch.randn(b, c, h // 2, w // 2, generator=g).to(dtype).float()
    y.backward(dy.double())
    assert rel(ops.avgpool_bwd(nhwc_rows(dy).to(dtype).to(DEV), b, h, w, c, 2), nhwc_rows(x64.grad)) < tol
    # attention-pool tokens
    hw = h * w
    pos = torch.randn(hw + 1, c, generator=g)
    xr = nhwc_rows(x).double().reshape(b, hw, c).requires_grad_(True)
    t = torch.cat([xr.mean(dim=1, keepdim=True), xr], dim=1) + pos.double()
    got = ops.attnpool_tokens_fwd(nhwc_rows(x).to(dtype).to(DEV), pos.to(DEV), b, hw)
    assert rel(got, t.reshape(-1, c)) < tol
    dt = torch.randn(b, hw + 1, c, generator=g).to(dtype).float()
    t.backward(dt.double())
    assert rel(ops.attnpool_tokens_bwd(dt.reshape(-1, c).to(dtype).to(DEV), b, hw), xr.grad.reshape(-1, c)) < tol


@pytest.mark.parametrize("relu,with_res", [(True, False), (False, False), (True, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,c", [(1000, 24), (1536, 16), (777, 64), (300, 2048)])
def test_batchnorm_train(ops, relu, with_res, dtype, rows, c):
    """Training-mode BatchNorm with fused ReLU / residual join against torch: output, input / residual / affine gradients, and the
    running statistics (momentum 0.1, unbiased variance); the two-part form (statistics of two half batches combined) gives the
    statistics of the whole batch - what a data-parallel run exchanges."""
    g = torch.Generator().manual_seed(3)      # c = 24: the general kernels; 16 / 64 / 2048: the four-channel-per-thread forms (1 / 1 / 2 channel chunks)
    x = (torch.randn(rows, c, generator=g) * 2 + 5).to(dtype).float()      # a mean far from zero: the shifted sums must not cancel
    res = torch.randn(rows, c, generator=g).to(dtype).float()
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    x64, r64 = x.double().requires_grad_(True), res.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    y = F.batch_norm(x64, rm, rv, g64, b64, training=True, momentum=0.1, eps=1e-5)
    if with_res:
        y = y + r64
    if relu:
        y = torch.relu(y)
    xd = x.to(dtype).to(DEV)
    stats = ops.bn_stats(xd)
    rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    mean, rstd = ops.bn_finish(stats, 1, c, rows, rmd, rvd)
    yd = ops.bn_apply(xd, mean, rstd, gamma.to(DEV), beta.to(DEV), relu, res.to(dtype).to(DEV) if with_res else None)
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    assert rel(yd, y) < tol
    assert rel(rmd, rm) < 1e-5 and rel(rvd, rv) < 1e-5
    # two half batches combined
    if rows % 2 == 0:
        s2 = torch.cat([ops.bn_stats(xd[: rows // 2].contiguous()), ops.bn_stats(xd[rows // 2:].contiguous())])
        mean2, rstd2 = ops.bn_finish(s2, 2, c, rows // 2)
        assert rel(mean2, mean) < 1e-6 and rel(rstd2, rstd) < 1e-5
    dy = torch.randn(rows, c, generator=g).to(dtype).float()
    y.backward(dy.double())
    dyd = dy.to(dtype).to(DEV)
    sums = ops.bn_bwd_stats(dyd, yd, xd, mean, rstd, relu)
    dg, db = torch.full((c,), 7.0, device=DEV), torch.full((c,), 7.0, device=DEV)
    dx, dres = ops.bn_bwd_apply(dyd, yd, xd, mean, rstd, gamma.to(DEV), sums, rows, relu, dg, db, False, want_dres=with_res)
    tolg = 1e-4 if dtype == torch.float32 else 3e-2
    assert rel(dx, x64.grad) < tolg
    assert rel(dg, g64.grad) < tolg and rel(db, b64.grad) < tolg
    if with_res:
        assert rel(dres, r64.grad) < tolg
    if relu and not with_res and ops.bn_mask_from_x(c):     # the ReLU mask recomputed from x instead of read from y: the same mask, the same bits
        sums2 = ops.bn_bwd_stats(dyd, None, xd, mean, rstd, relu, gamma.to(DEV), beta.to(DEV))
        dg2, db2 = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dx2, _ = ops.bn_bwd_apply(dyd, None, xd, mean, rstd, gamma.to(DEV), sums2, rows, relu, dg2, db2, False, beta=beta.to(DEV))
        assert torch.equal(sums2, sums) and torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)


def _pair(name, precision, seed=3):
    from oracle.clip_model import create_model
    from sparsify_clip_amd.model import ClipModel
    ref = create_model(name, seed=seed)
    model = ClipModel(name, device=DEV, precision=precision, seed=0)
    model.load_state_dict(ref.state_dict())
    return ref, model


@pytest.mark.parametrize("name,precision,im2col", [("test-rn", "fp32", False), ("test-rn64", "bf16", False), ("test-rn32", "bf16", False),
                                                   ("test-rn32", "bf16", True)])
def test_resnet_tower_forward_backward(ops, name, precision, im2col, monkeypatch):
    """The HIP ModifiedResNet tower against the oracle on identical weights: embeddings, every parameter gradient, the BatchNorm
    running statistics after the step; then evaluation mode (running statistics) against the oracle's eval forward."""
    from oracle.clip_model import synthetic_batch
    if im2col:      # the patch-matrix form of the 3x3 convolutions (SC_RN_IM2COL=1, the A/B switch) instead of the implicit GEMMs
        monkeypatch.setenv("SC_RN_IM2COL", "1")
    ref, model = _pair(name, precision)
    assert any(model.rn.implicit(cv) for cv in model.rn.convs()) == (precision == "bf16" and not im2col)
    # the zero-initialised last BatchNorm weights would hide every main-path gradient: give them values
    sd = ref.state_dict()
    gen = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith("bn3.weight") and "layer" in k:
            sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
    ref = ref.double()      # the oracle in fp64: in fp32 its own rounding (3-5e-6 on these gradients, measured during development) would be the yardstick
    ref.load_state_dict(sd)
    model.load_state_dict(sd)
    batch = 6
    images_np, _ = synthetic_batch(31, batch, ref.cfg)
    images = torch.tensor(images_np)
    ref.train()
    want = ref.encode_image(images.double())
    d_emb = torch.randn(batch, ref.cfg["embed_dim"], generator=gen)
    want.backward(d_emb.double())
    model.train()
    model.zero_grad()
    got = model.image_forward(images.to(DEV))
    model.image_backward(d_emb.to(DEV))
    torch.cuda.synchronize()
    # bf16 yardstick: the same network under torch's own CPU bf16 autocast.  This small random network amplifies rounding (fp32 torch
    # is already 1e-3 off the fp64 gradients in the stem), so "close to fp64" is not a usable bar for bf16; "as close as torch's bf16" is.
    auto = None
    if precision == "bf16":
        from oracle.clip_model import create_model
        rb = create_model(name, seed=3)
        rb.load_state_dict(sd)
        rb.train()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            wb = rb.encode_image(images)
        wb.float().backward(d_emb)
        auto = (wb.float(), {k: p.grad for k, p in rb.visual.named_parameters()})
        assert rel(got, want) < 1.5 * rel(auto[0], want) + 1e-2
    else:
        assert rel(got, want) < 1e-5
    # fp32 bar: 2e-2, not 1e-5.  Every kernel is pinned tightly on its own above; through the whole tower a single pre-activation within
    # rounding of zero takes the other side of the ReLU than in the fp64 oracle, and in the small late layers (24 .. 384 rows) one flipped
    # element moves a BatchNorm bias gradient - and everything upstream of it - by 1/rows: torch's own fp32 run is 1e-3 .. 5e-3 off the
    # fp64 gradients on this network for the same reason (measured during development).  The 1e-4 per-step loss bar is kept by the trajectory test.
    worst = ("", 0.0, 0.0)
    for k, p in ref.visual.named_parameters():
        if p.grad is None:
            continue
        if k == "attnpool.k_proj.bias":     # a bias on the keys shifts every score of a query alike: softmax cancels it, the gradient is rounding noise
            assert float(p.grad.abs().max()) < 1e-5 and float(model.grad("visual." + k).abs().max()) < (1e-5 if precision == "fp32" else 1e-2)
            continue
        e = rel(model.grad("visual." + k), p.grad)
        bar = 2e-2 if auto is None else 1.5 * rel(auto[1][k], p.grad) + 2e-2
        if e / bar > worst[1]:
            worst = (k, e / bar, e)
    assert worst[1] < 1.0, worst
    te = 1e-5 if precision == "fp32" else 6e-2
    for k, b in ref.visual.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(model.buffers["visual." + k]) == int(b)
        else:
            assert rel(model.buffers["visual." + k], b) < (1e-4 if precision == "fp32" else 2e-2), k
    ref.eval()
    model.eval()
    with torch.no_grad():
        want_e = ref.encode_image(images.double())
    assert rel(model.image_forward(images.to(DEV)), want_e) < te
    # state_dict round trip keeps the open_clip key set (parameters and buffers)
    keys = set(model.state_dict().keys())
    assert keys == set(ref.state_dict().keys()), keys ^ set(ref.state_dict().keys())


def test_resnet_training_trajectory_fp32(ops):
    """Whole training steps (both towers, experiment-6 loss stack, AdamW) on the ResNet model: per-step loss within 1e-4 of the
    oracle's CPU steps - the bar north_star sets for the ViT towers, applied to the model the reference YAMLs actually name."""
    from conftest import load_json
    from oracle.clip_model import create_model, synthetic_batch
    from oracle.train_step import CpuTrainer
    from sparsify_clip_amd.config import finalize_config
    from sparsify_clip_amd.model import ClipModel
    from sparsify_clip_amd.train import Trainer
    cfgs = load_json("configs.json")
    raw = cfgs[[k for k in cfgs if "experiment_6-" in k][0]]
    cfg = finalize_config(raw, 0, {"model": "test-rn", "batch_size": 8, "precision": "fp32"})
    ref_model = create_model("test-rn", seed=4)
    model = ClipModel("test-rn", device=DEV, precision="fp32", seed=0)
    model.load_state_dict(ref_model.state_dict())
    tr = Trainer(cfg, DEV, 10, model=model)
    oracle = CpuTrainer(cfg, 10, model=ref_model)
    tr.epoch = oracle.epoch = 1
    for k in range(4):
        images_np, tokens_np = synthetic_batch(500 + k, 8, ref_model.cfg)
        images, tokens = torch.tensor(images_np), torch.tensor(tokens_np)
        want = oracle.step(images, tokens).item()
        got = tr.step(images.to(DEV), tokens.to(DEV)).item()
        assert abs(got - want) <= 1e-4 * abs(want), (k, got, want)


@pytest.mark.parametrize("precision,batch", [("bf16", 64), ("fp32", 16)])
def test_rn50_reference_yaml_runs_unchanged(ops, precision, batch):
    """A reference YAML as it is (`model: "RN50"`, only the batch size reduced): the real-size ModifiedResNet-50 + text tower take
    three training steps; losses finite, parameters move, BatchNorm statistics updated, two runs bit-identical."""
    from conftest import load_json
    from sparsify_clip_amd.config import finalize_config
    from sparsify_clip_amd.data import synthetic_batch
    from sparsify_clip_amd.train import Trainer
    cfgs = load_json("configs.json")
    raw = cfgs[[k for k in cfgs if "experiment_6-" in k][0]]
    assert raw["model"] == "RN50"
    cfg = finalize_config(raw, 0, {"batch_size": batch, "precision": precision})
    images, tokens = [t.to(DEV) for t in synthetic_batch(7, batch)]
    runs = []
    for _ in range(2):
        tr = Trainer(cfg, DEV, 100)
        tr.epoch = 1
        p0 = tr.model.flat.clone()
        losses = [tr.step(images, tokens).item() for _ in range(3)]
        torch.cuda.synchronize()
        assert all(np.isfinite(l) for l in losses) and torch.isfinite(tr.model.flat).all()
        assert not torch.equal(p0, tr.model.flat)
        assert int(tr.model.buffers["visual.bn1.num_batches_tracked"]) == 3
        runs.append((losses, tr.model.flat.clone()))
        del tr
        torch.cuda.empty_cache()
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("b,h,w,cin,n", [(3, 6, 10, 64, 64), (2, 7, 7, 128, 256), (5, 14, 14, 64, 136), (1, 4, 4, 256, 128)])
def test_conv3x3_implicit_gemm(ops, b, h, w, cin, n):
    """sc_conv3x3_bf16 (implicit GEMM over a zero-bordered NHWC activation, no patch matrix) == F.conv2d(padding=1); with the weight
    re-arranged for the transposed convolution and the bordered output gradient as input it is the input gradient."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b, cin, h, w, generator=g).to(torch.bfloat16).float()
    wt = (torch.randn(n, cin, 3, 3, generator=g) * 0.05).to(torch.bfloat16).float()
    x64 = x.double().requires_grad_(True)
    y = F.conv2d(x64, wt.double(), padding=1)
    halo = F.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous().to(torch.bfloat16).to(DEV)          # [b, h+2, w+2, cin]
    w_taps = wt.permute(0, 2, 3, 1).reshape(n, 9 * cin).contiguous().to(torch.bfloat16).to(DEV)
    got = ops.conv3x3_bf16(halo, w_taps, b, h, w, out_dtype=torch.float32)
    assert rel(got, nhwc_rows(y)) < 1e-5
    dy = torch.randn(b, n, h, w, generator=g).to(torch.bfloat16).float()
    y.backward(dy.double())
    if n % 64 == 0 and (n // 64) & (n // 64 - 1) == 0:
        dhalo = F.pad(dy.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous().to(torch.bfloat16).to(DEV)     # [b, h+2, w+2, n]
        w_dx = wt.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * n).contiguous().to(torch.bfloat16).to(DEV)  # [cin][(8 - tap) * n + co]
        dx = ops.conv3x3_bf16(dhalo, w_dx, b, h, w, out_dtype=torch.float32)
        assert rel(dx, nhwc_rows(x64.grad)) < 1e-5


@pytest.mark.parametrize("b,h,w,c,cs", [(2, 7, 7, 64, 64), (3, 5, 9, 128, 128), (1, 4, 4, 2048, 2048), (3, 6, 5, 32, 64)])
def test_batchnorm_bordered_outputs(ops, b, h, w, c, cs):
    """sc_bn_apply / sc_bn_bwd_apply writing into the bordered image the implicit convolution reads: the interior carries the same bits as the
    compact output, the one-pixel border, the slack rows and the channels beyond c (a 32-channel activation in a 64-channel image) stay zero."""
    g = torch.Generator().manual_seed(11)
    rows = b * h * w
    x = (torch.randn(rows, c, generator=g) + 0.5).to(torch.bfloat16).to(DEV)
    dy = torch.randn(rows, c, generator=g).to(torch.bfloat16).to(DEV)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    mean, rstd = ops.bn_finish(ops.bn_stats(x), 1, c, rows)
    y = ops.bn_apply(x, mean, rstd, gamma, beta, True)
    flat, img = ops.halo_buffer(b, h, w, cs, torch.bfloat16, DEV)
    ops.bn_apply(x, mean, rstd, gamma, beta, True, halo=(img, h, w))
    want = torch.zeros(b, h + 2, w + 2, cs, dtype=torch.bfloat16, device=DEV)
    want[:, 1:-1, 1:-1, :c] = y.view(b, h, w, c)
    assert torch.equal(img.view(b, h + 2, w + 2, cs), want)
    assert not flat[: w + 3].any() and not flat[-(w + 3):].any()
    sums = ops.bn_bwd_stats(dy, None, x, mean, rstd, True, gamma, beta)
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dx, _ = ops.bn_bwd_apply(dy, None, x, mean, rstd, gamma, sums, rows, True, dg, db, False, beta=beta)
    flat2, img2 = ops.halo_buffer(b, h, w, cs, torch.bfloat16, DEV)
    dg2, db2 = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    ops.bn_bwd_apply(dy, None, x, mean, rstd, gamma, sums, rows, True, dg2, db2, False, beta=beta, halo=(img2, h, w))
    want[:, 1:-1, 1:-1, :c] = dx.view(b, h, w, c)
    assert torch.equal(img2.view(b, h + 2, w + 2, cs), want)
    assert torch.equal(dg2, dg) and torch.equal(db2, db)


@pytest.mark.parametrize("b,h,w,cin,cout", [(3, 6, 10, 64, 64), (2, 7, 7, 128, 256), (64, 14, 14, 64, 136), (1, 4, 4, 256, 128), (64, 8, 8, 512, 512)])
def test_conv3x3_implicit_weight_gradient(ops, b, h, w, cin, cout):
    """sc_conv3x3_dw_bf16 (one TN GEMM over the bordered images, the tap shift in the B operand's addressing) == the weight gradient of
    F.conv2d(padding=1); covers the small kernel (ragged contraction length) and the large one (rows a multiple of 64, >= 6 tiles)."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(b, cin, h, w, generator=g).to(torch.bfloat16).float()
    dz = (torch.randn(b, cout, h, w, generator=g) * 0.1).to(torch.bfloat16).float()
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(dz.double())
    xflat, ximg = ops.halo_buffer(b, h, w, cin, torch.bfloat16, DEV)
    ximg.view(b, h + 2, w + 2, cin)[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1).to(torch.bfloat16).to(DEV)
    _, dimg = ops.halo_buffer(b, h, w, cout, torch.bfloat16, DEV)
    dimg.view(b, h + 2, w + 2, cout)[:, 1:-1, 1:-1] = dz.permute(0, 2, 3, 1).to(torch.bfloat16).to(DEV)
    got = ops.conv3x3_dw_bf16(dimg, xflat, b, h, w).view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    assert rel(got, wt.grad) < 1e-5
    acc = torch.ones(cout, 9 * cin, device=DEV)
    ops.conv3x3_dw_bf16(dimg, xflat, b, h, w, out=acc, beta=1.0)
    assert rel(acc.view(cout, 3, 3, cin).permute(0, 3, 1, 2), wt.grad + 1) < 1e-5


def test_conv_entry_points_refuse_bad_shapes(ops):
    """The implicit-convolution entry points and their host wrappers fail loudly (status + message, no launch) on operands that do not match
    the kernels' assumptions: a channel count that is not 64 * 2^j, a bordered image of the wrong size, a weight matrix of the wrong width, a
    BatchNorm image narrower than the activation, a weight-gradient input without its slack rows."""
    from sparsify_clip_amd._lib import ScError
    b, h, w = 2, 6, 6
    with pytest.raises(ScError, match="64 times a power of two"):
        _, img = ops.halo_buffer(b, h, w, 192, torch.bfloat16, DEV)
        ops.conv3x3_bf16(img, torch.zeros(64, 9 * 192, dtype=torch.bfloat16, device=DEV), b, h, w)
    flat, img = ops.halo_buffer(b, h, w, 64, torch.bfloat16, DEV)
    wt = torch.zeros(64, 9 * 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ScError, match="a_halo has"):
        ops.conv3x3_bf16(img, wt, b, h + 1, w)
    with pytest.raises(ScError, match="w_taps is"):
        ops.conv3x3_bf16(img, wt[:, :-64].contiguous(), b, h, w)
    with pytest.raises(ScError, match="x_flat"):
        ops.conv3x3_dw_bf16(img, img, b, h, w)              # the image view instead of the flat buffer: no slack rows
    x = torch.zeros(b * h * w, 128, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.zeros(128, device=DEV), torch.ones(128, device=DEV)
    with pytest.raises(ScError, match="bordered image"):
        ops.bn_apply(x, mean, rstd, rstd, mean, True, halo=(img, h, w))      # a 128-channel activation into a 64-channel image
    assert ops.conv3x3_dw_bf16(img, flat, b, h, w).abs().max().item() == 0.0
