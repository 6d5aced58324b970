"""CPU tests of the host side: schedules, dispatch table, config loader, tokenizer/data, the C-ABI binding
(library loads and exports every symbol the header declares; no compute without a GPU), and loud failure of the
product path when it is asked to run on the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_json


def test_schedules_match_reference_tables():
    from sparsify_clip_amd import schedules as S
    sched = load_json("schedules.json")
    for row in sched["beta"]:
        assert [S.get_beta(s, row["total"], row["warmup"], row["ramp"]) for s in row["steps"]] == row["values"]
    for row in sched["alpha"]:
        assert [S.get_alpha(s, row["total"], row["warmup"], row["ramp"]) for s in row["steps"]] == row["values"]
    for row in sched["lr"]:
        lam = S.lr_lambda_factory(row["warmup_steps"], row["total"], config={"only_lunif_epochs": row["only_lunif_epochs"]})
        assert [lam(s) for s in row["steps"]] == row["values"]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = S.get_cosine_schedule_with_warmup(opt, 280, 1400, config={"only_lunif_epochs": 0})
    assert sch.get_last_lr()[0] == 0.0   # first optimiser step runs at lr = 0


def test_dispatch_table_covers_every_reference_yaml():
    from sparsify_clip_amd.config import finalize_config
    from sparsify_clip_amd.loss_dispatch import LOSS_TABLE, validate_loss_type
    from sparsify_clip_amd._lib import ScError
    from oracle.dispatch import TABLE
    cfgs = load_json("configs.json")
    assert len(cfgs) == 14 and sum(v is None for v in cfgs.values()) == 1     # 13 runnable YAMLs + the empty all_experiments.yaml
    seen = set()
    for rel, raw in cfgs.items():
        if raw is None:
            continue
        assert raw["model"] == "RN50"
        assert finalize_config(raw, 3)["model"] == "RN50"          # the YAMLs run as they are: ModifiedResNet is implemented (resnet.py)
        with pytest.raises(ScError, match="not implemented natively"):
            finalize_config(raw, 3, {"model": "RN101"})
        cfg = finalize_config(raw, 3, {"model": "ViT-B-32"})
        assert cfg["device_id"] == 3 and isinstance(cfg["learning_rate"], float) and cfg["learning_rate"] == 1e-4
        assert cfg["precision"] == "bf16" and cfg["model"] == "ViT-B-32" and cfg["batch_size"] == 256
        seen.add(cfg["loss_type"])
    assert seen == set(LOSS_TABLE) and len(seen) == 9
    for name, warm, unif, use_beta, use_alpha, use_lalign in TABLE:      # package table == oracle table (pinned by dispatch.json)
        spec = LOSS_TABLE[name]
        assert (spec.warmup_phase, spec.unif, spec.use_beta, spec.use_alpha, spec.use_lalign) == (warm, unif, use_beta, use_alpha, use_lalign)
    with pytest.raises(ScError):
        validate_loss_type("anchor-roberta")


def test_config_loader_files_dirs_overrides(tmp_path):
    from sparsify_clip_amd.config import config_files, load_config
    from sparsify_clip_amd._lib import ScError
    import yaml
    cfgs = load_json("configs.json")
    key = [k for k in cfgs if "experiment_10" in k][0]
    raw = dict(cfgs[key], learning_rate="1e-4")     # PyYAML parses 1e-4 as a string
    d = tmp_path / "cfgs"
    d.mkdir()
    (d / "b.yaml").write_text(yaml.safe_dump(raw))
    (d / "a_empty.yaml").write_text("# TODO\n")
    (d / "notes.txt").write_text("x")
    files = config_files(str(d))
    assert [os.path.basename(f) for f in files] == ["a_empty.yaml", "b.yaml"]
    assert load_config(files[0]) is None
    cfg = load_config(files[1], 1, {"model": "ViT-B-32", "batch_size": 4096, "precision": None})
    assert cfg["model"] == "ViT-B-32" and cfg["batch_size"] == 4096 and cfg["learning_rate"] == 1e-4 and cfg["alpha_warmup_epoch"] == 50
    bad = dict(raw)
    del bad["beta_decay_epoch"]
    (d / "c.yaml").write_text(yaml.safe_dump(bad))
    with pytest.raises(ScError):
        load_config(str(d / "c.yaml"))
    with pytest.raises(ScError):
        config_files(str(tmp_path / "missing"))


def test_cli_parses_reference_flags():
    import sparsify_clip as cli
    a = cli.parse_args(["--config", "x.yaml", "--device", "2"])
    assert a.config == "x.yaml" and a.device == 2 and a.model is None and a.batch_size is None
    with pytest.raises(SystemExit):
        cli.parse_args(["--config", "x.yaml"])          # --device is required, as in the reference
    for name in ["contrastive_loss", "lunif_loss", "lalign_loss", "sparsify_loss", "compute_centroids_only", "get_beta", "get_alpha",
                 "get_cosine_schedule_with_warmup", "train_model", "evaluate_model", "main", "set_seed"]:
        assert callable(getattr(cli, name))


def test_tokenizer_and_synthetic_data():
    from sparsify_clip_amd.data import EOT, SOT, HashTokenizer, SyntheticLoader, synthetic_batch
    tok = HashTokenizer()
    t = tok(["a dog on a bench", "A  DOG on a bench", " ".join(["w"] * 100)])
    assert t.shape == (3, 77) and t.dtype == torch.long
    assert torch.equal(t[0], t[1]) and t[0, 0] == SOT and t[0, 6] == EOT and not t[0, 7:].any()
    assert t[2, 0] == SOT and t[2, 76] == EOT                       # truncation keeps EOT last
    assert (t.argmax(-1) == torch.tensor([6, 6, 76])).all()
    images, tokens = synthetic_batch(42, 5)
    images2, tokens2 = synthetic_batch(42, 5)
    assert torch.equal(images, images2) and torch.equal(tokens, tokens2) and images.shape == (5, 3, 224, 224)
    eot_pos = tokens.argmax(-1)
    assert ((eot_pos >= 6) & (eot_pos <= 31)).all() and (tokens[:, 0] == SOT).all()
    loader = SyntheticLoader(40, 16, 1, "cpu", image_size=32, ctx=16, vocab=512)
    assert len(loader) == 2 and len(list(loader)) == 2             # drop_last semantics


def test_library_exports_every_declared_symbol():
    from sparsify_clip_amd import _lib
    header = open(os.path.join(ROOT, "include", "sparsify_hip.h")).read()
    declared = set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", header))
    declared -= {"sc_gemm_epilogue", "sc_block_desc"}
    assert declared == set(_lib.LIB.protos), declared ^ set(_lib.LIB.protos)
    dll = _lib.LIB.load()                      # raises if a declared symbol is not exported
    assert dll.sc_abi_version() == int(re.search(r"#define\s+SC_ABI_VERSION\s+(\d+)", header).group(1))
    assert dll.sc_abi_sizeof(0) == ctypes.sizeof(_lib.BlockDesc)
    assert dll.sc_abi_sizeof(1) == ctypes.sizeof(_lib.GemmEpilogue)
    assert 0 < dll.sc_loss_workspace_bytes(8192, 512) < 8192 * 8192 * 4   # the fused loss head keeps no [B,B] matrix
    # host-side argument validation works without a GPU and reports through sc_last_error
    rc = dll.sc_gemm_bf16_nt(128, 128, 100, None, 100, None, 100, None, 128, 1, None, None)
    assert rc < 0 and b"sc_gemm_bf16_nt" in dll.sc_last_error()
    rc = dll.sc_lunif_fwd_bwd(None, 1, 512, 2.0, 1.0, None, None, None, 0, None)
    assert rc < 0


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors or a CPU device raise instead of silently computing somewhere else."""
    from sparsify_clip_amd import losses
    from sparsify_clip_amd._lib import ScError
    from sparsify_clip_amd.model import ClipModel
    x = torch.randn(8, 16)
    for fn in (lambda: losses.contrastive_loss(x, x, 0.1), lambda: losses.lunif_loss(x), lambda: losses.lalign_loss(x, x),
               lambda: losses.sparsify_loss(x), lambda: losses.normalized_centroids(x, x)):
        with pytest.raises(ScError):
            fn()
    with pytest.raises(ScError):
        ClipModel("tiny", device="cpu")
    with pytest.raises(ScError):
        ClipModel("RN101", device="cuda:0")
    import sparsify_clip_amd
    src = "".join(open(os.path.join(ROOT, "sparsify_clip_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "sparsify_clip_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src       # the product never routes through the oracle


def test_uniformity_signatures_on_cpu(golden_metrics):
    import uniformity as U          # root-level alias, as `from uniformity import ...` in the reference
    from sparsify_clip_amd import uniformity as PU
    arr, v = golden_metrics
    f1, f2 = torch.tensor(arr["f1"]), torch.tensor(arr["f2"])
    assert abs(U.numpy_uniformity(f1, f2) - v["numpy_uniformity"]) < 1e-5 and isinstance(U.numpy_uniformity(f1, f2), float)
    assert abs(U.torch_uniformity(f1, f2).item() - v["torch_uniformity"]) < 1e-5
    assert abs(U.torch_uniformity1(f1).item() - v["torch_uniformity1"]) < 1e-5
    assert abs(U.torch_uniformity_equivalent(f1).item() - v["torch_uniformity_equivalent"]) < 1e-5
    assert abs(U.uniformity10(f1).item() - v["uniformity10"]) < 1e-4
    assert abs(PU.uniformity(f1, f2) - v["sparsify_clip.uniformity"]) < 1e-5
    assert abs(PU.compute_gap(f1, f2) - v["compute_gap"]) < 1e-6
    assert abs(PU.compute_mean_angular_value_of_a_modality(f1) - v["mean_angular_value_f1"]) < 1e-6
    assert abs(PU.mean_distance_of_true_pairs(f1, f2) - v["mean_distance_of_true_pairs"]) < 1e-6


def test_model_layout_without_gpu():
    """Parameter inventory of the native model == the oracle's open_clip-named state_dict (names, shapes, counts)."""
    from oracle.clip_model import create_model
    from sparsify_clip_amd import model as M
    from sparsify_clip_amd.resnet import ResNetVisual
    for name, total in [("ViT-B-32", 151_277_313), ("tiny", None), ("RN50", 102_007_137), ("test-rn", None)]:
        ref = create_model(name)
        stub = M.ClipModel.__new__(M.ClipModel)
        stub.cfg = M.CONFIGS[name]
        stub.rn = None
        if stub.cfg.get("v_kind") == "resnet":     # ModifiedResNet: parameters in the flat buffer, BatchNorm statistics as buffers
            stub.rn = ResNetVisual(stub, stub.cfg)
        else:
            stub.grid = stub.cfg["image_size"] // stub.cfg["patch"]
            stub.k_patch = 3 * stub.cfg["patch"] ** 2
        stub._layout()
        ref_shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        mine = {k: tuple(s) for k, (o, s) in stub.slots.items()}
        if stub.rn is not None:
            mine.update({k: tuple(shape) for k, shape, _, _ in stub.rn.buffer_specs()})
        assert mine == ref_shapes
        if total:
            assert sum(p.numel() for p in ref.parameters()) == total
        # buckets tile the trainable range without gaps or overlap, in backward order
        spans = [s for _, s in stub.buckets]
        assert spans[0][0] == 0 and spans[-1][1] == stub.n_trainable
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(off % 64 == 0 for off, _ in stub.slots.values())


def test_asm_audit_flags_the_three_hazards():
    """The build-time audit of the inline-assembly GEMM kernels (_asm_check.py) on hand-written snippets: a clean loop passes;
    compiler AGPR traffic, scratch, and a VALU instruction on a register whose inline-asm LDS read is still in flight are flagged;
    a counted lgkmcnt wait retires the older reads only."""
    from sparsify_clip_amd._asm_check import audit

    def kernel(body):
        return ".globl _Z23gemm_bf16_nt_big_kernelv\n_Z23gemm_bf16_nt_big_kernelv:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n"

    asm_read = lambda dst, addr="v9": f"\t;;#ASMSTART\n\tds_read_b128 {dst}, {addr} offset:0\n\t;;#ASMEND\n"
    mfma = "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_bf16 a[0:3], v[10:13], v[20:23], a[0:3]\n\t;;#ASMEND\n"
    clean = asm_read("v[10:13]") + asm_read("v[20:23]") + "\ts_waitcnt lgkmcnt(0)\n" + mfma + "\tv_add_u32_e32 v10, 1, v10\n"
    assert audit(kernel(clean)) == []
    early_use = asm_read("v[10:13]") + "\tv_add_u32_e32 v11, 1, v11\n\ts_waitcnt lgkmcnt(0)\n" + mfma
    assert any("still in flight" in p for p in audit(kernel(early_use)))
    assert any("AGPR traffic" in p for p in audit(kernel(clean + "\tv_accvgpr_write_b32 a5, v3\n")))
    assert any("scratch" in p for p in audit(kernel(clean + "\tscratch_store_dword off, v3, off\n")))
    # two reads, lgkmcnt(1): the older one (v[10:13]) has landed, the younger (v[20:23]) has not
    counted = asm_read("v[10:13]") + asm_read("v[20:23]") + "\ts_waitcnt lgkmcnt(1)\n"
    assert audit(kernel(counted + "\tv_add_u32_e32 v10, 1, v10\n\ts_waitcnt lgkmcnt(0)\n")) == []
    assert any("still in flight" in p for p in audit(kernel(counted + "\tv_add_u32_e32 v20, 1, v20\n\ts_waitcnt lgkmcnt(0)\n")))
    # a read in flight across a loop back-edge reaches code after the loop unless that code starts with a wait
    loop = ".LBB0_1:\n" + "\ts_waitcnt lgkmcnt(0)\n" + mfma + asm_read("v[10:13]") + "\ts_cbranch_scc1 .LBB0_1\n"
    assert any("still in flight" in p for p in audit(kernel(loop + "\tv_mov_b32_e32 v10, 0\n")))
    assert audit(kernel(loop + "\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v10, 0\n")) == []
    # other kernels in the same file are not audited
    assert audit(".globl _Z5otherv\n_Z5otherv:\n\tscratch_store_dword off, v3, off\n\ts_endpgm\n.Lfunc_end1:\n") == []


def test_input_pipeline_host_side(tmp_path):
    """Host logic of the input stage: the crop-box sampler == the oracle's restatement of torchvision's get_params (same draws from
    one Philox stream, boxes inside the image, the centre-crop fallback), and the COCO captions reader (sorted ids, Subset limit,
    captions per image) on a fixture in the reference's directory layout."""
    import json
    from PIL import Image
    from oracle import input_pipeline as O
    from sparsify_clip_amd import input_pipeline as IP
    for seed, (h, w) in enumerate([(480, 640), (640, 480), (50, 1000), (1000, 40), (7, 9)]):
        a = np.random.Generator(np.random.Philox(seed))
        b = np.random.Generator(np.random.Philox(seed))
        for _ in range(50):
            box = IP.random_resized_crop_params(a, h, w)
            assert box == O.random_resized_crop_params(b, h, w)
            top, left, bh, bw = box
            assert 0 <= top and 0 <= left and 0 < bh and 0 < bw and top + bh <= h and left + bw <= w
    # an impossible scale forces the fallback: whole image when the aspect ratio is admissible, clamped otherwise
    rng = np.random.Generator(np.random.Philox(0))
    assert IP.random_resized_crop_params(rng, 100, 120, scale=(4.0, 5.0)) == (0, 0, 100, 120)
    assert IP.random_resized_crop_params(rng, 100, 400, scale=(4.0, 5.0)) == (0, (400 - 133) // 2, 100, 133)
    assert IP.MEAN == O.MEAN and IP.STD == O.STD
    root = tmp_path / "imgs"
    root.mkdir()
    images, anns = [], []
    for iid in (30, 10, 20):
        Image.fromarray(np.full((4 + iid // 10, 6, 3), iid, dtype=np.uint8), "RGB").save(root / f"{iid}.png")
        images.append({"id": iid, "file_name": f"{iid}.png"})
        anns += [{"image_id": iid, "id": iid * 10 + c, "caption": f"c{c}-{iid}"} for c in range(3)]
    ann = tmp_path / "captions.json"
    ann.write_text(json.dumps({"images": images, "annotations": anns}))
    ds = IP.CocoCaptionsDataset(str(root), str(ann))
    assert len(ds) == 3 and [int(ds[i][0][0, 0, 0]) for i in range(3)] == [10, 20, 30]
    assert ds[1][1] == ["c0-20", "c1-20", "c2-20"] and ds[2][0].shape == (7, 6, 3) and ds[0][0].dtype == np.uint8
    assert len(IP.CocoCaptionsDataset(str(root), str(ann), limit=2)) == 2
    syn = IP.SyntheticCocoDataset(10, seed=1, pool=4)
    img, caps = syn[7]
    assert img.dtype == np.uint8 and img.ndim == 3 and len(caps) == 5 and syn[7][1] == caps


def test_host_gather_rows_native():
    """sc_host_gather_rows (a HOST function of the library: C++ threads, no GPU call) packs strided crop views into one contiguous
    staging buffer exactly as numpy would, for any thread count; items that would overrun the buffer are refused."""
    import ctypes
    from sparsify_clip_amd._lib import LIB, ScError
    rng = np.random.Generator(np.random.Philox(5))
    imgs = [rng.integers(0, 256, size=(int(rng.integers(20, 60)), int(rng.integers(20, 60)), 3), dtype=np.uint8) for _ in range(37)]
    crops = []
    for im in imgs:
        h, w = im.shape[:2]
        bh, bw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
        t, l = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
        crops.append(im[t:t + bh, l:l + bw])
    want = np.concatenate([np.ascontiguousarray(c).reshape(-1) for c in crops])
    sizes = np.array([c.size for c in crops], dtype=np.int64)
    offset = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    n = len(crops)
    src = (ctypes.c_void_p * n)(*[c.ctypes.data for c in crops])
    stride = np.array([c.strides[0] for c in crops], dtype=np.int64)
    rows = np.array([c.shape[0] for c in crops], dtype=np.int64)
    row_bytes = np.array([c.shape[1] * 3 for c in crops], dtype=np.int64)
    for threads in (1, 3, 16):
        dst = np.zeros(want.size + 7, dtype=np.uint8)
        LIB.call("sc_host_gather_rows", n, src, stride.ctypes.data, rows.ctypes.data, row_bytes.ctypes.data, dst.ctypes.data, offset.ctypes.data,
                 dst.size, threads)
        assert np.array_equal(dst[: want.size], want) and not dst[want.size:].any()
    with pytest.raises(ScError):
        LIB.call("sc_host_gather_rows", n, src, stride.ctypes.data, rows.ctypes.data, row_bytes.ctypes.data, dst.ctypes.data, offset.ctypes.data,
                 want.size - 1, 2)
    LIB.call("sc_host_gather_rows", 0, None, None, None, None, None, None, 0, 4)      # an empty batch is a no-op


def test_bench_simulated_world_patches_and_restores_dist():
    """bench.py's N = 1 'one rank of a w-GPU job' measurement patches the four meeting points of the trainer with sparsify_clip_amd.dist
    for its duration only (the simulation does not live in the product module)."""
    import bench
    from sparsify_clip_amd import dist as D
    before = (D.all_gather_embeddings, D.sharding, D.exchange_packets, D.local_rows)
    img, txt = torch.nn.functional.normalize(torch.randn(4, 8), dim=-1), torch.nn.functional.normalize(torch.randn(4, 8), dim=-1)
    assert D.sharding() == (1, 0) and D.all_gather_embeddings(img, txt)[0] is img
    with bench.simulated_world(4):
        gi, gt = D.all_gather_embeddings(img, txt)
        assert gi.shape == (16, 8) and torch.equal(gi[:4], img) and torch.equal(gt[:4], txt)
        assert torch.allclose(gi.norm(dim=-1), torch.ones(16), atol=1e-6)
        assert D.sharding() == (4, 0)
        assert D.exchange_packets(torch.arange(3.0)).shape == (4, 3)
        assert torch.equal(D.local_rows(gi), img)
    assert (D.all_gather_embeddings, D.sharding, D.exchange_packets, D.local_rows) == before
    assert D.sharding() == (1, 0) and D.local_rows(img) is img
