"""GPU parity tests, kernel level: every C-ABI entry point against a CPU reference (fp64 where cheap).

Tolerances: fp32 kernels (MFMA f32 = fmaf chain) rel 1e-5..1e-4 on values; bf16 kernels are compared with the
same computation on bf16-ROUNDED inputs in fp64, so only accumulation order and the output rounding differ.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sparsify_clip_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def assert_close(got, want, rtol, atol, what=""):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        idx = bad.nonzero()[:5].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} elements off, max abs err {err.max():.3e} "
                             f"(want max {want.abs().max():.3e}), first bad idx {idx}, "
                             f"got {[got[tuple(i)].item() for i in idx]} want {[want[tuple(i)].item() for i in idx]}, "
                             f"nonfinite {int((~torch.isfinite(got)).sum())}")


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / np.sqrt(2.0)))


def gelu_grad64(x):
    return 0.5 * (1 + torch.erf(x / np.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


# ------------------------------------------------------------------------------------------------ fp32 GEMM
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(200, 136, 75), (256, 384, 512), (33, 515, 1030)])
def test_gemm_f32_orientations(ops, ta, tb, m, n, k):
    a = rnd(k, m, seed=1) if ta else rnd(m, k, seed=1)
    b = rnd(n, k, seed=2) if tb else rnd(k, n, seed=2)
    want = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    got = ops.gemm_f32(a.to(DEV), b.to(DEV), trans_a=bool(ta), trans_b=bool(tb))
    assert_close(got, want, 1e-5, 1e-5 * np.sqrt(k), f"gemm_f32 ta={ta} tb={tb}")


def test_gemm_f32_epilogue(ops):
    m, n, k = 130, 260, 96
    a, w, bias, resid, pre = rnd(m, k, seed=3), rnd(n, k, seed=4), rnd(n, seed=5), rnd(m, n, seed=6), rnd(m, n, seed=7)
    acc = a.double() @ w.double().t()
    # forward-style: bias + gelu + residual, pre-activation kept
    pre_out = torch.empty(m, n, device=DEV)
    e = ops.make_epilogue(alpha=0.5, bias=bias.to(DEV), pre_out=pre_out, act=1, resid=resid.to(DEV), ld_aux=n)
    got = ops.gemm_f32(a.to(DEV), w.to(DEV), trans_b=True, epi=e)
    v = 0.5 * acc + bias.double()
    assert_close(pre_out, v, 1e-5, 1e-4, "pre_out")
    assert_close(got, gelu64(v) + resid.double(), 1e-5, 1e-4, "bias+gelu+resid")
    # backward-style: multiply by gelu'(pre), accumulate into C
    c0 = rnd(m, n, seed=8)
    out = c0.to(DEV).clone()
    e = ops.make_epilogue(beta=1.0, dgelu_pre=pre.to(DEV), ld_aux=n)
    ops.gemm_f32(a.to(DEV), w.to(DEV), trans_b=True, out=out, epi=e)
    assert_close(out, acc * gelu_grad64(pre.double()) + c0.double(), 1e-5, 1e-4, "dgelu+beta")


# ------------------------------------------------------------------------------------------------ bf16 GEMMs
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (300, 192, 128), (1024, 768, 768), (77, 2304, 512)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_nt(ops, m, n, k, out_dtype):
    a, b = rnd(m, k, seed=11, dtype=torch.bfloat16), rnd(n, k, seed=12, dtype=torch.bfloat16)
    want = a.double() @ b.double().t()
    got = ops.gemm_bf16_nt(a.to(DEV), b.to(DEV), out_dtype=out_dtype)
    rt = 1e-2 if out_dtype == torch.bfloat16 else 1e-4
    assert_close(got, want, rt, 2e-3 * np.sqrt(k), f"gemm_bf16_nt {m}x{n}x{k}")


@pytest.mark.parametrize("m,n,k", [(4096, 1536, 64), (4100, 1544, 192), (4352, 2304, 768), (5000, 1800, 128), (10300, 2056, 128), (33000, 520, 64),
                                   (32768, 1024, 192), (49152, 768, 64), (16384, 512, 128), (8320, 768, 128)])
def test_gemm_bf16_nt_wide(ops, m, n, k):
    """Shapes that dispatch to the 256x256 AGPR kernels (M >= 4096, N >= 512): full and ragged tiles, odd / even K-tile counts, the
    4th-6th with more than 256 tiles so that the leftover rows go to a second launch of the 256x128 kernel; the whole-tile shapes
    run the PERSISTENT kernel for the step's four epilogues (512 whole tiles = 2 per workgroup; 510 whole + 132 half tiles; 256 half
    tiles only; 96 whole tiles + a last 128-row panel of 3 half tiles) and the one-tile-per-workgroup kernel for the rest; every epilogue, checked element by element against fp64 on the same
    bf16 operands (computed on the device)."""
    a, w = rnd(m, k, seed=21, dtype=torch.bfloat16).to(DEV), rnd(n, k, seed=22, scale=0.1, dtype=torch.bfloat16).to(DEV)
    bias, resid = rnd(n, seed=23).to(DEV), rnd(m, n, seed=24).to(DEV)
    acc = a.double() @ w.double().t()
    got = ops.gemm_bf16_nt(a, w, out_dtype=torch.float32)
    assert_close(got, acc, 1e-4, 2e-3 * np.sqrt(k), "wide plain fp32")
    pre_out = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    e = ops.make_epilogue(bias=bias, pre_out=pre_out, act=1, ld_aux=n)
    got = ops.gemm_bf16_nt(a, w, epi=e)
    v = acc + bias.double()
    assert_close(pre_out, v, 1e-2, 1e-2, "wide pre_out")
    assert_close(got, gelu64(v), 1e-2, 1e-2, "wide gelu")
    e = ops.make_epilogue(bias=bias, resid=resid, ld_aux=n)
    got = ops.gemm_bf16_nt(a, w, out_dtype=torch.float32, epi=e)
    assert_close(got, v + resid.double(), 1e-4, 2e-3, "wide fp32 out + resid")
    pre = rnd(m, n, seed=25, dtype=torch.bfloat16).to(DEV)
    e = ops.make_epilogue(dgelu_pre=pre, ld_aux=n)
    got = ops.gemm_bf16_nt(a, w, epi=e)
    assert_close(got, acc * gelu_grad64(pre.double()), 1e-2, 1e-2, "wide dgelu")
    got2 = ops.gemm_bf16_nt(a, w, epi=e)
    assert torch.equal(got, got2), "not bit-stable run to run"
    cs = torch.full((n,), 3.0, device=DEV)          # fused column sums of the stored output (c_fc bias gradient), accumulate form
    e = ops.make_epilogue(dgelu_pre=pre, ld_aux=n, colsum=cs, colsum_accumulate=True, rows=m)
    got3 = ops.gemm_bf16_nt(a, w, epi=e)
    assert torch.equal(got3, got)
    assert_close(cs, got.double().sum(0) + 3.0, 1e-4, 1e-3 * np.sqrt(m), "wide dgelu colsum")
    got = ops.gemm_bf16_nt(a, w, epi=ops.make_epilogue(bias=bias, ld_aux=n))
    assert_close(got, v, 1e-2, 1e-2, "wide bias")


@pytest.mark.parametrize("m,n,k", [(65536, 512, 192), (43520, 768, 256), (8192, 2048, 256), (12416, 768, 192)])
def test_gemm_bf16_nt_tile_tickets(ops, m, n, k):
    """Dynamic tile order of the persistent NT kernel (sc_gemm_epilogue.tile_tickets): whichever workgroup computes a tile, the stored
    values are the ones of the fixed tile lists - bit for bit, for the step's four epilogues, with whole and half tiles - the
    tickets are zero again after every launch, and a launch that shares the GPU with another stream's kernel (a diagnostic kernel
    keeping 48 CUs) still computes every tile exactly once."""
    import ctypes
    from sparsify_clip_amd._lib import LIB
    a, w = rnd(m, k, seed=31, dtype=torch.bfloat16).to(DEV), rnd(n, k, seed=32, scale=0.1, dtype=torch.bfloat16).to(DEV)
    bias, resid = rnd(n, seed=33).to(DEV), rnd(m, n, seed=34).to(DEV)
    pre = rnd(m, n, seed=35, dtype=torch.bfloat16).to(DEV)
    tk = torch.zeros(16, dtype=torch.int32, device=DEV)
    occupy = LIB.load().sc_debug_occupy
    occupy.argtypes, occupy.restype = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p], ctypes.c_int
    side = torch.cuda.Stream()

    def run(kind, tickets, taken=0):
        cs = torch.zeros(n, device=DEV)
        pre_out = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
        kw = dict(tile_tickets=tickets)
        if kind == "bias":
            e, od = ops.make_epilogue(bias=bias, ld_aux=n, **kw), torch.bfloat16
        elif kind == "gelu":
            e, od = ops.make_epilogue(bias=bias, pre_out=pre_out, act=1, ld_aux=n, **kw), torch.bfloat16
        elif kind == "resid":
            e, od = ops.make_epilogue(bias=bias, resid=resid, ld_aux=n, **kw), torch.float32
        else:
            e, od = ops.make_epilogue(dgelu_pre=pre, ld_aux=n, colsum=cs, rows=m, **kw), torch.bfloat16
        if taken:
            with torch.cuda.stream(side):
                assert occupy(taken, 1500, ctypes.c_void_p(side.cuda_stream)) == 0
            torch.cuda._sleep(100000)
        out = ops.gemm_bf16_nt(a, w, out_dtype=od, epi=e)
        torch.cuda.synchronize()
        return out, pre_out, cs

    for kind in ("bias", "gelu", "resid", "dgelu"):
        ref = run(kind, None)
        for taken in (0, 48):
            got = run(kind, tk, taken)
            assert torch.equal(got[0], ref[0]), f"{kind}: output differs with tile tickets ({taken} CUs taken)"
            if kind == "gelu":
                assert torch.equal(got[1], ref[1]), "gelu: pre-activation differs with tile tickets"
            if kind == "dgelu":
                assert torch.equal(got[2], ref[2]), "dgelu: fused column sums differ with tile tickets"
            assert int(tk.abs().sum()) == 0, f"{kind}: tickets not left zero"


def test_gemm_bf16_epilogue_gelu_series_accuracy(ops):
    """The bf16 epilogues evaluate GELU / GELU' as odd polynomial series (csrc/common.h; degree 6 / 7, fitted by
    tools/fit_gelu_series.py); through an fp32-output GEMM with acc == x (B = I) resp. acc == 1 the series are compared with the erf
    forms over [-9, 9].  The step rounds both results to bf16 (half an ulp = 2^-8 = 3.9e-3 relative), so the bounds are stated as
    fractions of that rounding: GELU within 7.5e-5 |x| everywhere (the error of Phi) and within 1.5e-4 relative for x > 0 (1/26 of
    the rounding); GELU' within 3e-4 absolute (1/13 of the rounding of a value near 1)."""
    m, k = 512, 64
    x = torch.linspace(-9, 9, m * k).reshape(m, k).to(torch.bfloat16)
    eye = torch.eye(k, dtype=torch.bfloat16)
    got = ops.gemm_bf16_nt(x.to(DEV), eye.to(DEV), out_dtype=torch.float32, epi=ops.make_epilogue(act=1, ld_aux=k)).double().cpu()
    want = gelu64(x.double())
    err = (got - want).abs()
    assert float((err - 7.5e-5 * x.double().abs()).max()) <= 1e-7, f"GELU series: max |err| / |x| = {float((err / x.double().abs().clamp_min(1e-3)).max()):.3e}"
    pos = x.double() > 0
    assert float((err[pos] / want[pos]).max()) <= 1.5e-4, "GELU series: relative error for x > 0"
    a = torch.zeros(m, k, dtype=torch.bfloat16); a[:, 0] = 1
    b = torch.zeros(k, k, dtype=torch.bfloat16); b[:, 0] = 1
    pre = x.to(DEV)
    got = ops.gemm_bf16_nt(a.to(DEV), b.to(DEV), out_dtype=torch.float32, epi=ops.make_epilogue(dgelu_pre=pre, ld_aux=k))
    assert_close(got, gelu_grad64(x.double()), 0, 3e-4, "GELU' series")


@pytest.mark.parametrize("m,n,k", [(4100, 1544, 192), (8192, 768, 128), (300, 192, 128), (10300, 2056, 64)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_nt_epilogue_colsum(ops, m, n, k, out_dtype):
    """Column sums of the stored output taken by the epilogue (256x256 kernel: fused; small problems: fallback pass): equal to
    summing the stored C, with and without accumulation, and the GEMM result itself is unchanged."""
    a, w = rnd(m, k, seed=41, dtype=torch.bfloat16).to(DEV), rnd(n, k, seed=42, scale=0.1, dtype=torch.bfloat16).to(DEV)
    pre = rnd(m, n, seed=43, dtype=torch.bfloat16).to(DEV)
    want_c = ops.gemm_bf16_nt(a, w, out_dtype=out_dtype, epi=ops.make_epilogue(dgelu_pre=pre, ld_aux=n))
    cs = torch.full((n,), 5.0, device=DEV)
    got_c = ops.gemm_bf16_nt(a, w, out_dtype=out_dtype, epi=ops.make_epilogue(dgelu_pre=pre, ld_aux=n, colsum=cs, rows=m))
    assert torch.equal(got_c, want_c), "the fused column sums must not change C"
    want_s = want_c.double().sum(0)
    assert_close(cs, want_s, 1e-5, 2e-4 * np.sqrt(m), "epilogue colsum")
    cs2 = torch.full((n,), 5.0, device=DEV)
    ops.gemm_bf16_nt(a, w, out_dtype=out_dtype, epi=ops.make_epilogue(dgelu_pre=pre, ld_aux=n, colsum=cs2, colsum_accumulate=True, rows=m))
    assert_close(cs2, want_s + 5.0, 1e-5, 2e-4 * np.sqrt(m), "epilogue colsum, accumulate")
    cs3 = torch.empty(n, device=DEV)
    ops.gemm_bf16_nt(a, w, out_dtype=out_dtype, epi=ops.make_epilogue(dgelu_pre=pre, ld_aux=n, colsum=cs3, rows=m))
    assert torch.equal(cs, cs3)


def test_gemm_bf16_nt_asymmetric_identity(ops):
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md section 3)."""
    n, k = 256, 128
    a = torch.eye(k, dtype=torch.bfloat16)
    b = (torch.arange(n)[:, None] * 1.0 + torch.arange(k)[None, :] * 0.25).to(torch.bfloat16)
    got = ops.gemm_bf16_nt(a.to(DEV), b.to(DEV), out_dtype=torch.float32)
    assert_close(got, b.double().t(), 0, 1e-6, "identity x B^T")


def test_gemm_bf16_nt_epilogue(ops):
    m, n, k = 200, 256, 192
    a, w = rnd(m, k, seed=13, dtype=torch.bfloat16), rnd(n, k, seed=14, scale=0.1, dtype=torch.bfloat16)
    bias, resid = rnd(n, seed=15), rnd(m, n, seed=16)
    acc = a.double() @ w.double().t()
    pre_out = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    e = ops.make_epilogue(bias=bias.to(DEV), pre_out=pre_out, act=1, ld_aux=n)
    got = ops.gemm_bf16_nt(a.to(DEV), w.to(DEV), epi=e)
    v = acc + bias.double()
    assert_close(pre_out, v, 1e-2, 1e-2, "bf16 pre_out")
    assert_close(got, gelu64(v), 1e-2, 1e-2, "bf16 gelu")
    e = ops.make_epilogue(bias=bias.to(DEV), resid=resid.to(DEV), ld_aux=n)
    got = ops.gemm_bf16_nt(a.to(DEV), w.to(DEV), out_dtype=torch.float32, epi=e)
    assert_close(got, v + resid.double(), 1e-4, 2e-3, "fp32 out + resid")
    pre = rnd(m, n, seed=17, dtype=torch.bfloat16)
    e = ops.make_epilogue(dgelu_pre=pre.to(DEV), ld_aux=n)
    got = ops.gemm_bf16_nt(a.to(DEV), w.to(DEV), epi=e)
    assert_close(got, acc * gelu_grad64(pre.double()), 1e-2, 1e-2, "bf16 dgelu")


@pytest.mark.parametrize("r,m,n", [(64, 128, 128), (200, 136, 264), (4096, 768, 768), (1600, 2304, 768), (640, 776, 520), (64, 1536, 512)])
def test_gemm_bf16_tn(ops, r, m, n):
    a, b = rnd(r, m, seed=21, dtype=torch.bfloat16), rnd(r, n, seed=22, dtype=torch.bfloat16)
    want = a.double().t() @ b.double()
    got = ops.gemm_bf16_tn(a.to(DEV), b.to(DEV))
    assert_close(got, want, 1e-4, 2e-3 * np.sqrt(r), f"gemm_bf16_tn r={r}")
    c0 = rnd(m, n, seed=23)
    out = c0.to(DEV).clone()
    ops.gemm_bf16_tn(a.to(DEV), b.to(DEV), out=out, alpha=0.5, beta=1.0)
    assert_close(out, 0.5 * want + c0.double(), 1e-4, 2e-3 * np.sqrt(r), "tn alpha/beta")


@pytest.mark.parametrize("r,m,n", [(64, 128, 128), (200, 136, 264), (4096, 768, 768), (1600, 2304, 768), (51200, 1536, 512), (640, 776, 520), (192, 1536, 512)])
def test_gemm_bf16_tn_fused_colsum(ops, r, m, n):
    """Bias gradient fused into the weight-gradient GEMM: column sums of A from the staged tiles, with and without a split
    contraction, ragged M, accumulate on/off; the GEMM result itself must not change."""
    a, b = rnd(r, m, seed=31, dtype=torch.bfloat16).to(DEV), rnd(r, n, seed=32, dtype=torch.bfloat16).to(DEV)
    want_c = a.double().t() @ b.double()
    want_s = a.double().sum(0)
    cs = torch.full((m,), 7.0, device=DEV)
    got = ops.gemm_bf16_tn(a, b, colsum_out=cs)
    assert_close(got, want_c, 1e-4, 2e-3 * np.sqrt(r), "tn + colsum: C")
    assert_close(cs, want_s, 1e-5, 1e-4 * np.sqrt(r), "tn + colsum: sums")
    assert torch.equal(got, ops.gemm_bf16_tn(a, b)), "fusing the column sums must not change C"
    cs2 = torch.full((m,), 3.0, device=DEV)
    ops.gemm_bf16_tn(a, b, colsum_out=cs2, colsum_beta=1.0)
    assert_close(cs2, want_s + 3.0, 1e-5, 1e-4 * np.sqrt(r), "tn + colsum: accumulate")
    cs3 = torch.empty(m, device=DEV)
    ops.gemm_bf16_tn(a, b, colsum_out=cs3)
    assert torch.equal(cs, cs3), "column sums are deterministic"


def test_gemm_bf16_tn_asymmetric(ops):
    r, m, n = 64, 128, 128
    a = torch.zeros(r, m, dtype=torch.bfloat16)
    a[torch.arange(r), torch.arange(r)] = 1.0          # A^T picks rows of B
    b = (torch.arange(r)[:, None] * 1.0 + torch.arange(n)[None, :] * 0.5).to(torch.bfloat16)
    got = ops.gemm_bf16_tn(a.to(DEV), b.to(DEV))
    want = a.double().t() @ b.double()
    assert_close(got, want, 0, 1e-6, "tn selector")


# ------------------------------------------------------------------------------------------------ loss head
def _oracle():
    from oracle import loss_head as L
    return L


@pytest.mark.parametrize("tag", ["rand32x512", "clustered64x768"])
def test_loss_head_small_golden(ops, golden_small, tag):
    arr, meta = golden_small
    v = meta[tag]["values"]
    img, txt = torch.tensor(arr[f"{tag}/img"]).to(DEV), torch.tensor(arr[f"{tag}/txt"]).to(DEV)
    R = 1e-4  # north_star: per-step loss within 1e-4 relative of the reference fp32 CPU path
    loss, di, dt, dtemp = ops.contrastive_fwd_bwd(img, txt, 0.1, need_dtemp=True)
    assert abs(loss.item() - v["contrastive_T0.1"]) <= R * abs(v["contrastive_T0.1"])
    assert_close(di, torch.tensor(arr[f"{tag}/contrastive_T0.1.g0"]), 1e-3, 1e-7, "contrastive d_img")
    assert_close(dt, torch.tensor(arr[f"{tag}/contrastive_T0.1.g1"]), 1e-3, 1e-7, "contrastive d_txt")
    assert abs(dtemp.item() - v["contrastive_learnableT.g2.value"]) <= 1e-3 * abs(v["contrastive_learnableT.g2.value"])
    loss, _, _, _ = ops.contrastive_fwd_bwd(img, txt, 0.07, need_grad=False)
    assert abs(loss.item() - v["contrastive_T0.07"]) <= R * abs(v["contrastive_T0.07"])
    loss, dx = ops.lunif_fwd_bwd(img, 2.0)
    assert abs(loss.item() - v["lunif_img"]) <= R * abs(v["lunif_img"])
    assert_close(dx, torch.tensor(arr[f"{tag}/lunif_img.g0"]), 2e-3, 2e-7, "lunif d_img")
    loss, dx = ops.lunif_fwd_bwd(txt, 3.0)
    assert abs(loss.item() - v["lunif_txt_t3"]) <= R * abs(v["lunif_txt_t3"])
    assert_close(dx, torch.tensor(arr[f"{tag}/lunif_txt_t3.g0"]), 2e-3, 2e-7, "lunif t=3 d_txt")
    loss, dx, dy = ops.lalign_fwd_bwd(img, txt, 2.0)
    assert abs(loss.item() - v["lalign"]) <= R * abs(v["lalign"])
    assert_close(dx, torch.tensor(arr[f"{tag}/lalign.g0"]), 1e-4, 1e-8, "lalign dx")
    assert_close(dy, torch.tensor(arr[f"{tag}/lalign.g1"]), 1e-4, 1e-8, "lalign dy")
    loss, dx, _ = ops.lalign_fwd_bwd(img, txt, 1.0)
    assert abs(loss.item() - v["lalign_alpha1"]) <= R * abs(v["lalign_alpha1"])
    assert_close(dx, torch.tensor(arr[f"{tag}/lalign_alpha1.g0"]), 1e-4, 1e-8, "lalign alpha=1 dx")
    loss, dx = ops.sparsify_fwd_bwd(img)
    assert abs(loss.item() - v["sparsify_img"]) <= R * abs(v["sparsify_img"])
    assert_close(dx, torch.tensor(arr[f"{tag}/sparsify_img.g0"]), 1e-3, 1e-7, "sparsify dx")
    # centroids: normalize((a+b)/2) -> lunif -> back to a and b
    c, inv = ops.centroid_fwd(img, txt)
    loss, dc = ops.lunif_fwd_bwd(c, 2.0)
    assert abs(loss.item() - v["lunif_centroids"]) <= R * abs(v["lunif_centroids"])
    da, db = torch.zeros_like(img), torch.zeros_like(txt)
    ops.centroid_bwd_accumulate(c, inv, dc, da, db)
    assert_close(da, torch.tensor(arr[f"{tag}/lunif_centroids.g0"]), 2e-3, 2e-7, "centroid d_img")
    assert_close(db, torch.tensor(arr[f"{tag}/lunif_centroids.g1"]), 2e-3, 2e-7, "centroid d_txt")


@pytest.mark.parametrize("key", ["b512_d512_rand", "b4096_d512_rand", "b8192_d512_rand", "b4096_d768_rand", "b2048_d512_clustered"])
def test_loss_head_large_golden(ops, golden_large, key):
    """BASELINE sizes: inputs regenerated from the Philox seed, reference outputs from the fixture."""
    L = _oracle()
    g = golden_large[key]
    v = g["values"]
    img_np, txt_np = L.philox_embeddings(g["seed"], g["b"], g["d"], g["clustered"])
    img, txt = torch.tensor(img_np).to(DEV), torch.tensor(txt_np).to(DEV)
    probe = [(i % g["b"], j % g["d"]) for i, j in g["probe_index"]]
    R = 1e-4

    def rel(a, b):
        return abs(a - b) <= R * abs(b)

    loss, di, dt, dtemp = ops.contrastive_fwd_bwd(img, txt, 0.1, need_dtemp=True)
    assert rel(loss.item(), v["contrastive_T0.1"]), (loss.item(), v["contrastive_T0.1"])
    assert abs(di.double().norm().item() - v["contrastive_T0.1.g0.norm"]) <= 1e-3 * v["contrastive_T0.1.g0.norm"]
    assert abs(dt.double().norm().item() - v["contrastive_T0.1.g1.norm"]) <= 1e-3 * v["contrastive_T0.1.g1.norm"]
    np.testing.assert_allclose([di[i, j].item() for i, j in probe], v["contrastive_T0.1.g0.probe"], rtol=5e-3, atol=1e-9)
    assert abs(dtemp.item() - v["contrastive_learnableT.g2.value"]) <= 2e-3 * abs(v["contrastive_learnableT.g2.value"])
    loss, dx = ops.lunif_fwd_bwd(img, 2.0)
    assert rel(loss.item(), v["lunif_img"]), (loss.item(), v["lunif_img"])
    assert abs(dx.double().norm().item() - v["lunif_img.g0.norm"]) <= 2e-3 * v["lunif_img.g0.norm"]
    np.testing.assert_allclose([dx[i, j].item() for i, j in probe], v["lunif_img.g0.probe"], rtol=1e-2, atol=1e-9)
    loss, dx, _ = ops.lalign_fwd_bwd(img, txt, 2.0)
    assert rel(loss.item(), v["lalign"])
    c, inv = ops.centroid_fwd(img, txt)
    loss, _ = ops.lunif_fwd_bwd(c, 2.0, need_grad=False)
    assert rel(loss.item(), v["lunif_centroids"]), (loss.item(), v["lunif_centroids"])
    loss, _ = ops.sparsify_fwd_bwd(img, need_grad=False)
    assert rel(loss.item(), v["sparsify_img"])
    # closed-form fp64 gradient of the oracle, full tensor (not only probes), at sizes it finishes in seconds
    if g["b"] <= 4096:
        _, d64, _, _ = L.contrastive_grads(img_np, txt_np, 0.1)
        assert_close(di, torch.tensor(d64), 2e-3, 1e-8, "contrastive d_img vs fp64")
        _, dx64 = L.lunif_grads(img_np, 2.0)
        _, dxg = ops.lunif_fwd_bwd(img, 2.0)
        assert_close(dxg, torch.tensor(dx64), 5e-3, 1e-8, "lunif dx vs fp64")


def test_loss_head_properties(ops):
    """Size-independent properties: permutation invariance, symmetry, determinism, identical-pair edge cases."""
    L = _oracle()
    img_np, txt_np = L.philox_embeddings(5, 1000, 512)       # ragged: not a multiple of any tile
    img, txt = torch.tensor(img_np).to(DEV), torch.tensor(txt_np).to(DEV)
    l1, d1, _, _ = ops.contrastive_fwd_bwd(img, txt, 0.1)
    l2, d2, _, _ = ops.contrastive_fwd_bwd(img, txt, 0.1)
    assert l1.item() == l2.item() and torch.equal(d1, d2), "not bit-stable run to run"
    lt, _, _, _ = ops.contrastive_fwd_bwd(txt, img, 0.1)
    assert abs(lt.item() - l1.item()) <= 1e-6 * abs(l1.item()), "InfoNCE must be symmetric in its arguments"
    perm = torch.randperm(1000, generator=torch.Generator().manual_seed(0)).to(DEV)
    lu, _ = ops.lunif_fwd_bwd(img, 2.0)
    lp, _ = ops.lunif_fwd_bwd(img[perm].contiguous(), 2.0)
    assert abs(lu.item() - lp.item()) <= 2e-6 * abs(lu.item())
    want = L.lunif_loss_gram(torch.tensor(img_np).double()).item()
    assert abs(lu.item() - want) <= 1e-5 * abs(want)
    # x == y: lalign is exactly 0 with a zero (sub-)gradient, as torch.norm's backward gives
    la, dx, dy = ops.lalign_fwd_bwd(img, img.clone(), 2.0)
    assert la.item() == 0.0 and not dx.any() and not dy.any()
    la, dx, _ = ops.lalign_fwd_bwd(img, img.clone(), 1.0)
    assert la.item() == 0.0 and torch.isfinite(dx).all() and not dx.any()
    # smallest batch
    l, d, _, _ = ops.contrastive_fwd_bwd(img[:2].contiguous(), txt[:2].contiguous(), 0.1)
    want, d64, _, _ = L.contrastive_grads(img_np[:2], txt_np[:2], 0.1)
    assert abs(l.item() - want) <= 1e-5 * abs(want)
    assert_close(d, torch.tensor(d64), 1e-4, 1e-7, "B=2 grad")
    from sparsify_clip_amd._lib import ScError
    with pytest.raises(ScError):
        ops.lunif_fwd_bwd(img[:1].contiguous(), 2.0)          # a single row has no pairs


def test_l2norm_and_axpy(ops):
    L = _oracle()
    x = rnd(37, 512, seed=31)
    y, inv = ops.l2norm_fwd(x.to(DEV), 0.0)
    assert_close(y, L.normalize_rows(x.double()), 1e-6, 1e-7, "l2norm fwd")
    dy = rnd(37, 512, seed=32)
    dx = ops.l2norm_bwd(y, inv, dy.to(DEV))
    assert_close(dx, torch.tensor(L.normalize_backward(x.numpy(), dy.numpy())), 1e-4, 1e-6, "l2norm bwd")
    a = rnd(1000, seed=33).to(DEV)
    b = rnd(1000, seed=34).to(DEV)
    want = b.double().cpu() + 0.25 * a.double().cpu()
    ops.axpy_(b, 0.25, a)
    assert_close(b, want, 1e-6, 1e-6, "axpy")


def test_retrieval_ranks(ops, golden_metrics):
    from oracle import metrics as M
    arr, v = golden_metrics
    f1, f2 = torch.tensor(arr["f1"]), torch.tensor(arr["f2"])
    score = f2 @ (0.6 * f2 + 0.4 * f1).t()
    rf, rb, tf, tb = ops.retrieval_ranks(score.to(DEV))
    assert rf.cpu().tolist() == M.retrieval_ranks(score, "forward").tolist()
    assert rb.cpu().tolist() == M.retrieval_ranks(score, "backward").tolist()
    assert tf.cpu().tolist() == v["top1_forward"]          # bit-exact top-1 indices (north_star)
    assert tb.cpu().tolist() == v["top1_backward"]


# ------------------------------------------------------------------------------------------------ encoder pieces
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,w", [(203, 512), (64, 768), (10, 1024)])
def test_layernorm(ops, dtype, rows, w):
    x, gam, bet = rnd(rows, w, seed=41, scale=2.0), 1 + 0.1 * rnd(w, seed=42), 0.1 * rnd(w, seed=43)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), gam.to(DEV), bet.to(DEV), dtype)
    x64 = x.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x64, (w,), g64, b64, 1e-5)
    tol = (1e-5, 1e-5) if dtype == torch.float32 else (1e-2, 1e-2)
    assert_close(y, ref, *tol, "ln fwd")
    dy = rnd(rows, w, seed=44).to(dtype)
    dres = rnd(rows, w, seed=45)
    ref.backward(dy.double())
    colsum = torch.ones(w, device=DEV)
    dx, dx_cast, dg, db = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gam.to(DEV), dres=dres.to(DEV), want_cast=True, dx_colsum=colsum)
    assert_close(dx, x64.grad + dres.double(), 1e-4, 1e-4, "ln dx")
    assert_close(colsum, (x64.grad + dres.double()).sum(0), 1e-4, 1e-3, "ln dx column sums (overwrite)")
    dx2, _, _, _ = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gam.to(DEV))
    assert_close(dx2, x64.grad, 1e-4, 1e-4, "ln dx without residual")
    assert_close(dx_cast, x64.grad + dres.double(), *tol if dtype == torch.bfloat16 else (1e-4, 1e-4), "ln dx cast")
    assert_close(dg, g64.grad, 1e-4, 1e-3, "ln dgamma")
    assert_close(db, b64.grad, 1e-4, 1e-3, "ln dbeta")


def _ref_attention(qkv, batch, seq, heads, causal):
    w = qkv.shape[1] // 3
    q, k, v = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in qkv.split(w, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((seq, seq), float("-inf"), dtype=s.dtype).triu_(1)
    o = torch.softmax(s, dim=-1) @ v
    return o.transpose(1, 2).reshape(batch * seq, w)


def _poison_lds():
    """Diagnostic hook of the library: fills every CU's LDS with NaN bit patterns (LDS is not cleared between kernels)."""
    import ctypes
    from sparsify_clip_amd._lib import LIB
    f = LIB.load().sc_debug_poison_lds
    f.argtypes, f.restype = [ctypes.c_void_p], ctypes.c_int
    assert f(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seq,heads,causal", [(50, 3, False), (77, 2, True), (16, 1, True), (5, 2, False), (128, 1, False), (128, 2, True), (100, 1, True),
                                              (90, 1, True), (257, 2, False), (200, 1, True), (129, 2, True), (145, 1, False)])
def test_attention(ops, dtype, seq, heads, causal):
    if seq > 128 and dtype == torch.float32:
        pytest.skip("the fp32 (parity-path) attention kernel keeps the whole head in LDS: S <= 128")
    batch, w = 3, heads * 64
    qkv = rnd(batch * seq, 3 * w, seed=51).to(dtype)
    q64 = qkv.double().requires_grad_(True)
    ref = _ref_attention(q64, batch, seq, heads, causal)
    _poison_lds()
    out = ops.attention_fwd(qkv.to(DEV), batch, seq, heads, causal)
    tol = (1e-5, 1e-5) if dtype == torch.float32 else (1e-2, 1e-2)
    assert_close(out, ref, *tol, "attention fwd")
    if dtype == torch.float32 and seq > 90:
        # the stand-alone fp32 (parity-path) backward keeps scores and dP of a head in LDS: S <= 90; inside sc_block_bwd longer
        # sequences are composed from the fp32 GEMM (tests/test_gpu_model.py: test-s101, test-l14 in fp32)
        from sparsify_clip_amd._lib import ScError
        with pytest.raises(ScError, match="LDS"):
            ops.attention_bwd(qkv.to(DEV), rnd(batch * seq, w, seed=52).to(DEV), batch, seq, heads, causal)
    else:      # every other case with a forward (round 1 skipped the backward at 77 < S <= 128)
        d_out = rnd(batch * seq, w, seed=52).to(dtype)
        ref.backward(d_out.double())
        _poison_lds()   # LDS the kernels do not write themselves reads as NaN (sequence lengths that leave whole tiles of the kernel's shape empty)
        d_qkv = ops.attention_bwd(qkv.to(DEV), d_out.to(DEV), batch, seq, heads, causal)
        tolb = (1e-4, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
        assert_close(d_qkv, q64.grad, *tolb, "attention bwd")
        if ops.attention_uses_stats(dtype, seq):
            # the flash-attention form (sc_attention_fwd_stats / _bwd_stats): same output, lse_i = -log2 sum_j exp(s_ij), and a backward that
            # takes the statistics and the forward's output instead of recomputing row maxima / sums (delta_i = dO_i . O_i)
            lse = torch.full((batch * heads * seq,), float("nan"), device=DEV)
            _poison_lds()
            out2 = ops.attention_fwd(qkv.to(DEV), batch, seq, heads, causal, lse=lse)
            assert torch.equal(out2, out), "the statistics forward must not change the output"
            w = heads * 64
            q, k, _ = [t.reshape(batch, seq, heads, 64).transpose(1, 2) for t in qkv.double().split(w, dim=1)]
            sc = q @ k.transpose(-1, -2) * 0.125
            if causal:
                sc = sc + torch.full((seq, seq), float("-inf"), dtype=sc.dtype).triu_(1)
            assert_close(lse.reshape(batch, heads, seq), -torch.logsumexp(sc, dim=-1) / np.log(2.0), 1e-4, 1e-4, "attention lse")
            _poison_lds()
            cs = torch.zeros(3 * w, device=DEV)
            d2 = ops.attention_bwd(qkv.to(DEV), d_out.to(DEV), batch, seq, heads, causal, colsum_out=cs, out=out2, lse=lse)
            assert_close(d2, q64.grad, *tolb, "attention bwd (statistics form)")
            assert_close(cs, d2.double().sum(0), 1e-5, 2e-4, "attention bwd colsum (statistics form)")
            d3 = ops.attention_bwd(qkv.to(DEV), d_out.to(DEV), batch, seq, heads, causal, out=out2, lse=lse)
            assert torch.equal(d3, d2), "statistics backward: not bit-stable with / without the column sums"


@pytest.mark.parametrize("seq,heads,causal", [(50, 12, False), (77, 8, True), (33, 5, True)])
def test_attention_bwd_many_heads(ops, seq, heads, causal):
    """The short-sequence backward is a persistent kernel (one workgroup per CU walks groups of four heads, the next group's operands
    prefetched into registers): with 110 images x `heads` heads every workgroup runs several groups, the last one partly filled.
    Against torch autograd in fp64 on the bf16-rounded inputs, with the fused in_proj bias gradient."""
    batch, w = 110, heads * 64
    qkv = rnd(batch * seq, 3 * w, seed=71).to(torch.bfloat16)
    d_out = rnd(batch * seq, w, seed=72).to(torch.bfloat16)
    q64 = qkv.double().requires_grad_(True)
    _ref_attention(q64, batch, seq, heads, causal).backward(d_out.double())
    _poison_lds()
    cs = torch.zeros(3 * w, device=DEV)
    got = ops.attention_bwd(qkv.to(DEV), d_out.to(DEV), batch, seq, heads, causal, colsum_out=cs)
    assert_close(got, q64.grad, 2e-2, 2e-2, "attention bwd, many heads")
    assert_close(cs, got.double().sum(0), 1e-5, 2e-4, "attention bwd colsum, many heads")
    again = ops.attention_bwd(qkv.to(DEV), d_out.to(DEV), batch, seq, heads, causal)
    assert torch.equal(again, got), "attention bwd: not bit-stable between launches / with and without the column sums"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seq,heads,causal", [(50, 3, False), (77, 2, True), (16, 1, True), (257, 2, False)])
def test_attention_bwd_fused_colsum(ops, dtype, seq, heads, causal):
    """in_proj bias gradient from the attention backward: equal to summing the stored d_qkv (bit-identical d_qkv, sums within fp32
    reassociation), on the one-wave and one-workgroup MFMA kernels and on the fallback pass of the fp32 path."""
    if seq > 128 and dtype == torch.float32:
        pytest.skip("fp32 attention: S <= 128")
    batch, w = 5, heads * 64
    qkv = rnd(batch * seq, 3 * w, seed=61).to(dtype).to(DEV)
    d_out = rnd(batch * seq, w, seed=62).to(dtype).to(DEV)
    want = ops.attention_bwd(qkv, d_out, batch, seq, heads, causal)
    cs = torch.full((3 * w,), 2.0, device=DEV)
    got = ops.attention_bwd(qkv, d_out, batch, seq, heads, causal, colsum_out=cs)
    assert torch.equal(got, want)
    assert_close(cs, want.double().sum(0), 1e-5, 1e-4, "attention bwd colsum")
    cs2 = torch.full((3 * w,), 2.0, device=DEV)
    ops.attention_bwd(qkv, d_out, batch, seq, heads, causal, colsum_out=cs2, colsum_accumulate=True)
    assert_close(cs2, want.double().sum(0) + 2.0, 1e-5, 1e-4, "attention bwd colsum, accumulate")


@pytest.mark.parametrize("batch,seq,vocab", [(37, 77, 49408), (5, 24, 1000), (1024, 77, 49408), (3, 8, 7)])
def test_token_sort(ops, batch, seq, vocab):
    """sc_token_sort (index bookkeeping of the token-embedding scatter-add): keys = token id up to the EOT, `vocab` behind it; the result is
    the stable sort of those keys - bit-equal to torch.sort(stable=True) on the same keys, for ragged captions and repeated tokens."""
    g = torch.Generator().manual_seed(batch * 1000 + seq)
    tokens = torch.randint(1, vocab - 1, (batch, seq), generator=g)
    lengths = torch.randint(1, seq + 1, (batch,), generator=g)
    for b in range(batch):
        tokens[b, lengths[b] - 1] = vocab - 1          # the EOT is the largest id (open_clip: argmax pooling)
        tokens[b, lengths[b]:] = 0
    eot = (lengths - 1).to(torch.int32)
    keys, order = ops.token_sort(tokens.to(DEV), eot.to(DEV), vocab)
    pos = torch.arange(seq)[None, :]
    want_keys = torch.where(pos <= eot[:, None], tokens, torch.full_like(tokens, vocab)).reshape(-1)
    wk, wo = torch.sort(want_keys, stable=True)
    assert torch.equal(keys.cpu(), wk) and torch.equal(order.cpu(), wo)


def test_colsum_cast_transpose(ops):
    x = rnd(1000, 2304, seed=61)
    assert_close(ops.colsum(x.to(DEV)), x.double().sum(0), 1e-5, 1e-3, "colsum f32")
    xb = x.to(torch.bfloat16)
    out = torch.ones(2304, device=DEV)
    ops.colsum(xb.to(DEV), out=out, accumulate=True)
    assert_close(out, xb.double().sum(0) + 1, 1e-5, 1e-3, "colsum bf16 accumulate")
    src = rnd(70, 130, seed=62)
    assert torch.equal(ops.cast_bf16(src.to(DEV)).cpu(), src.to(torch.bfloat16))
    assert torch.equal(ops.transpose_cast_bf16(src.to(DEV)).cpu(), src.t().contiguous().to(torch.bfloat16))
    # the batched form (one launch over a pointer table): ragged sizes, every item
    srcs = [rnd(r, c, seed=63 + k).to(DEV) for k, (r, c) in enumerate([(70, 130), (32, 32), (1, 97), (257, 64), (768, 3072)])]
    dsts = [torch.zeros(t.shape[1], t.shape[0], dtype=torch.bfloat16, device=DEV) for t in srcs]
    run = ops.transpose_cast_bf16_batch(list(zip(srcs, dsts)))
    run()
    for t, d in zip(srcs, dsts):
        assert torch.equal(d, t.t().contiguous().to(torch.bfloat16))


@pytest.mark.parametrize("res,p", [(64, 32), (64, 16), (56, 14)])      # 32 / 16: the vectorised compile-time-P kernel; 14 (ViT-L/14): the generic one
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vit_stem(ops, dtype, res, p):
    b, w = 3, 128
    g = res // p
    img = rnd(b, 3, res, res, seed=71)
    cols = ops.im2col(img.to(DEV), p, (3 * p * p + 64) // 64 * 64, dtype)
    ref = torch.nn.functional.unfold(img, kernel_size=p, stride=p).transpose(1, 2).reshape(b * g * g, 3 * p * p)
    assert torch.equal(cols[:, : 3 * p * p].float().cpu(), ref.to(dtype).float())
    assert not cols[:, 3 * p * p:].any()
    seq = g * g + 1
    patch_out, cls, pos = rnd(b * g * g, w, seed=72).to(dtype), rnd(w, seed=73), rnd(seq, w, seed=74)
    x = ops.vit_tokens_fwd(patch_out.to(DEV), cls.to(DEV), pos.to(DEV), b, seq)
    want = torch.cat([cls.expand(b, 1, w), patch_out.float().reshape(b, g * g, w)], dim=1) + pos
    assert_close(x, want.reshape(b * seq, w), 1e-6, 1e-6, "vit tokens fwd")
    dx = rnd(b * seq, w, seed=75)
    d_cls, d_pos = torch.ones(w, device=DEV), torch.ones(seq, w, device=DEV)
    d_patch = ops.vit_tokens_bwd(dx.to(DEV), b, seq, dtype, d_cls, d_pos, True)
    dx3 = dx.reshape(b, seq, w)
    assert torch.equal(d_patch.float().cpu(), dx3[:, 1:].reshape(-1, w).to(dtype).float())
    assert_close(d_cls, dx3[:, 0].double().sum(0) + 1, 1e-6, 1e-5, "d_cls")
    assert_close(d_pos, dx3.double().sum(0) + 1, 1e-6, 1e-5, "d_pos")


def test_text_stem_and_pool(ops):
    b, s, w, vocab = 5, 16, 64, 300
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(1, vocab - 2, (b, s), generator=g)
    tokens[:, 0] = vocab - 2
    eot_pos = torch.tensor([5, 9, 3, 15, 7])
    for i in range(b):
        tokens[i, eot_pos[i]] = vocab - 1
        tokens[i, eot_pos[i] + 1:] = 0
    tokens[1, 1] = tokens[0, 1]  # a repeated word across captions
    emb, pos = rnd(vocab, w, seed=81), rnd(s, w, seed=82)
    x = ops.text_embed_fwd(tokens.to(DEV), emb.to(DEV), pos.to(DEV))
    assert_close(x, (emb[tokens] + pos).reshape(b * s, w), 0, 0, "text embed fwd")
    eot = ops.argmax_tokens(tokens.to(DEV))
    assert eot.cpu().tolist() == eot_pos.tolist() == tokens.argmax(-1).tolist()
    pooled = ops.pool_gather(x, eot, b, s)
    assert torch.equal(pooled.cpu(), (emb[tokens] + pos)[torch.arange(b), eot_pos])
    first = ops.pool_gather(x, None, b, s)
    assert torch.equal(first.cpu(), (emb[tokens] + pos)[:, 0])
    # backward: rows after EOT carry zero gradient and are left out of the sorted list
    dx = rnd(b * s, w, seed=83).reshape(b, s, w)
    active = torch.arange(s)[None, :] <= eot_pos[:, None]
    dx = (dx * active[..., None]).reshape(b * s, w)
    flat = tokens.reshape(-1)
    idx = active.reshape(-1).nonzero().squeeze(1)
    st, perm = torch.sort(flat[idx], stable=True)
    order = idx[perm]
    d_emb, d_pos = torch.zeros(vocab, w, device=DEV), torch.zeros(s, w, device=DEV)
    ops.text_embed_bwd(dx.to(DEV), st.to(DEV), order.to(DEV), b, s, d_emb, d_pos, False)
    want = torch.zeros(vocab, w, dtype=torch.float64).index_add_(0, flat, dx.double())
    assert_close(d_emb, want, 1e-6, 1e-6, "token embedding scatter-add")
    assert_close(d_pos, dx.reshape(b, s, w).double().sum(0), 1e-6, 1e-6, "text d_pos")
    dxs = torch.zeros(b * s, w, device=DEV)
    ops.pool_scatter(pooled, eot, b, s, dxs)
    want = torch.zeros(b, s, w)
    want[torch.arange(b), eot_pos] = pooled.cpu()
    assert torch.equal(dxs.cpu(), want.reshape(b * s, w))


@pytest.mark.parametrize("batch,seq,width,vocab", [(300, 24, 512, 50), (129, 16, 768, 9), (64, 8, 256, 3), (1024, 32, 512, 49408), (70, 77, 1024, 400)])
def test_token_scatter_long_runs(ops, batch, seq, width, vocab):
    """Token-embedding gradient with runs of equal tokens far longer than 64 sorted positions (start / end of text: one per caption; tiny
    vocabularies): runs inside a block of 64 sorted positions belong to token_scatter_kernel, runs that cross a boundary to
    token_scatter_long_kernel's workgroup at the first boundary crossed - also when the run starts exactly on a boundary or covers whole
    blocks.  Against an fp64 index_add; bit-stable run to run; the accumulate form adds."""
    g = torch.Generator().manual_seed(batch + seq)
    tokens = torch.randint(1, max(2, vocab - 2), (batch, seq), generator=g)
    tokens[:, 0] = vocab - 2
    lengths = torch.randint(2, seq + 1, (batch,), generator=g)
    for b in range(batch):
        tokens[b, lengths[b] - 1] = vocab - 1
        tokens[b, lengths[b]:] = 0
    eot = (lengths - 1).to(torch.int32)
    emb, pos = rnd(vocab, width, seed=84), rnd(seq, width, seed=85)
    x = ops.text_embed_fwd(tokens.to(DEV), emb.to(DEV), pos.to(DEV))
    assert torch.equal(x.cpu(), (emb[tokens] + pos).reshape(batch * seq, width))
    active = torch.arange(seq)[None, :] <= eot[:, None]
    dx = (rnd(batch * seq, width, seed=86).reshape(batch, seq, width) * active[..., None]).reshape(batch * seq, width)
    keys, order = ops.token_sort(tokens.to(DEV), eot.to(DEV), vocab)
    d_emb, d_pos = torch.full((vocab, width), 7.0, device=DEV), torch.zeros(seq, width, device=DEV)
    ops.text_embed_bwd(dx.to(DEV), keys, order, batch, seq, d_emb, d_pos, False)
    want = torch.zeros(vocab, width, dtype=torch.float64).index_add_(0, tokens.reshape(-1), dx.double())
    tol = 1e-5 * float(np.sqrt(batch))       # fp32 sums of up to `batch` terms of unit size (the start / end-of-text rows) against fp64
    assert_close(d_emb, want, 1e-5, tol, "token embedding scatter-add, long runs")
    again = torch.zeros(vocab, width, device=DEV)
    ops.text_embed_bwd(dx.to(DEV), keys, order, batch, seq, again, torch.zeros(seq, width, device=DEV), False)
    assert torch.equal(again, d_emb), "not bit-stable"
    ops.text_embed_bwd(dx.to(DEV), keys, order, batch, seq, again, torch.zeros(seq, width, device=DEV), True)
    assert_close(again, 2 * want, 1e-5, 2 * tol, "token embedding scatter-add, accumulate")


def test_adamw_matches_torch(ops):
    n = 5003
    p0, g = rnd(n, seed=91), [rnd(n, seed=92 + i, scale=0.1) for i in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3)
    p = p0.to(DEV).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step, gi in enumerate(g, 1):
        ref.grad = gi.clone()
        opt.step()
        ops.adamw_step(p, gi.to(DEV), m, v, shadow, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
    assert_close(p, ref.detach(), 1e-6, 1e-7, "adamw params")
    assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))


def _variant_worker(env_key, env_val, q):
    """Child process: the env knob is read once per process by the library, so each variant needs a fresh one."""
    import os
    os.environ[env_key] = env_val
    import numpy as np
    import torch
    from sparsify_clip_amd import ops
    g = torch.Generator().manual_seed(5)
    bad = []
    for m, n, k in [(4352, 768, 768), (4100, 1544, 192)]:
        a = (torch.randn(m, k, generator=g)).to(torch.bfloat16).cuda()
        w = (torch.randn(n, k, generator=g) * 0.1).to(torch.bfloat16).cuda()
        bias, resid = torch.randn(n, generator=g).cuda(), torch.randn(m, n, generator=g).cuda()
        acc = a.double() @ w.double().t()
        got = ops.gemm_bf16_nt(a, w, out_dtype=torch.float32, epi=ops.make_epilogue(bias=bias, resid=resid, ld_aux=n))
        err = (got.double() - (acc + bias.double() + resid.double())).abs().max().item()
        if not err <= 5e-3:
            bad.append(("nt", m, n, k, err))
        dy = torch.randn(m, n, generator=g).to(torch.bfloat16).cuda()
        cs = torch.zeros(n, device="cuda")
        dw = ops.gemm_bf16_tn(dy, a, colsum_out=cs)
        want = dy.double().t() @ a.double()
        err = ((dw.double() - want).abs().max() / want.abs().max()).item()
        if not err <= 1e-4:
            bad.append(("tn", m, n, k, err))
        err = (cs.double() - dy.double().sum(0)).abs().max().item()
        if not err <= 1e-2:
            bad.append(("tn colsum", m, n, k, err))
    q.put(bad)


@pytest.mark.parametrize("env_key,env_val", [("SC_GEMM_NT", "2"), ("SC_GEMM_NT", "1"), ("SC_GEMM_TN", "128")])
def test_gemm_kernel_variants_shipped_in_the_library(ops, env_key, env_val):
    """The non-default GEMM kernels that remain in the .so (256x128 3-stage NT everywhere, 128x128 NT, 128x128 TN everywhere) run
    the wide shapes correctly too - one child process per knob value (the library reads its knobs once per process)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_variant_worker, args=(env_key, env_val, q))
    p.start()
    bad = q.get(timeout=600)
    p.join(60)
    assert p.exitcode == 0 and bad == [], (env_key, env_val, bad)


@pytest.mark.parametrize("r,w", [(4096, 256), (5120, 512), (1024, 768)])
def test_gemm_bf16_tn_group(ops, r, w):
    """The four weight gradients of a block in one launch (sc_gemm_bf16_tn_group): each equals the single-problem TN GEMM result
    within fp32 reassociation (the contraction is split differently), overwrite and accumulate, bit-stable run to run."""
    shapes = [(w, 4 * w), (4 * w, w), (w, w), (3 * w, w)]
    probs, want = [], []
    for k, (m, n) in enumerate(shapes):
        a = rnd(r, m, seed=200 + k, dtype=torch.bfloat16).to(DEV)
        b = rnd(r, n, seed=210 + k, dtype=torch.bfloat16).to(DEV)
        c = torch.full((m, n), 0.5, device=DEV)
        probs.append((a, b, c))
        want.append(a.double().t() @ b.double())
    ops.gemm_bf16_tn_group(probs, beta=1.0)
    for (a, b, c), w64 in zip(probs, want):
        assert_close(c, w64 + 0.5, 1e-4, 2e-3 * np.sqrt(r) * 0.05, "grouped dW (accumulate)")
    first = [p[2].clone() for p in probs]
    ops.gemm_bf16_tn_group(probs, beta=0.0)
    for (a, b, c), w64 in zip(probs, want):
        assert_close(c, w64, 1e-4, 2e-3 * np.sqrt(r) * 0.05, "grouped dW (overwrite)")
    again = [p[2].clone() for p in probs]
    ops.gemm_bf16_tn_group(probs, beta=0.0)
    assert all(torch.equal(x, p[2]) for x, p in zip(again, probs))
    assert all(not torch.equal(x, y) for x, y in zip(first, again))
    ops.gemm_bf16_tn_group(probs[:2], beta=0.0)       # fewer than four problems
    assert_close(probs[1][2], want[1], 1e-4, 2e-3 * np.sqrt(r) * 0.05, "two problems")
