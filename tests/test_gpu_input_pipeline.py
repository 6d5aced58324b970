"""GPU parity tests of the input stage (reference sparsify_clip.py:992-1065): the device resample / flip /
normalise kernels against the reference's own arithmetic - torchvision's functional transforms restated on the REAL Pillow resampler
(oracle/input_pipeline.py) - bit for bit, and the loader end to end (synthetic COCO-like data, a COCO-format fixture directory)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device(DEV)


def _device_batch(gpu, crops, flips, size):
    """crops: list of uint8 [h,w,3] arrays (already cut out) -> fp32 [n,3,size,size] through the C ABI."""
    from sparsify_clip_amd import ops
    from sparsify_clip_amd.input_pipeline import MEAN, STD
    sizes = np.array([c.size for c in crops], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    dims = np.array([[c.shape[0], c.shape[1], 0, 0, c.shape[0], c.shape[1]] for c in crops], dtype=np.int32)
    tsz = dims[:, 4].astype(np.int64) * size * 3
    toff = np.concatenate([[0], np.cumsum(tsz)[:-1]]).astype(np.int64)
    pix = torch.from_numpy(np.concatenate([np.ascontiguousarray(c).reshape(-1) for c in crops])).to(gpu)
    t = lambda a: torch.from_numpy(a).to(gpu)
    return ops.image_resample_normalize(pix, t(off), t(dims.reshape(-1)), t(np.array(flips, dtype=np.int32)), t(toff), len(crops), int(dims[:, 4].max()),
                                        int(tsz.sum()), size, MEAN, STD)


def test_resample_normalize_bit_exact_vs_pillow(gpu):
    """Down-scales (2.9x, 6x, 12x), up-scales, 1:1, extreme aspect ratios, a one-pixel-wide crop; with and without flip."""
    from oracle.input_pipeline import resized_crop_normalize
    rng = np.random.Generator(np.random.Philox(11))
    shapes = [(480, 640), (640, 480), (224, 224), (37, 53), (1, 300), (300, 1), (1344, 2688), (100, 1000), (223, 225), (2700, 90)]
    crops, flips, want = [], [], []
    for k, (h, w) in enumerate(shapes):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        if k % 3 == 0:     # smooth content as well as noise
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], axis=-1).astype(np.uint8)
        flip = bool(k % 2)
        crops.append(img), flips.append(int(flip))
        want.append(resized_crop_normalize(img, (0, 0, h, w), flip, 224))
    got = _device_batch(gpu, crops, flips, 224).cpu()
    for k, w_ in enumerate(want):
        assert torch.equal(got[k], w_), (shapes[k], (got[k] - w_).abs().max().item())


def test_loader_train_and_eval_match_the_cpu_pipeline(gpu):
    """DeviceAugLoader on the synthetic COCO-like dataset: every image of a train batch (random crop + flip) and of an eval batch
    (Resize) equals the oracle's output for the same box / flip; two loaders with one seed give identical batches; one random
    caption of the five per image."""
    from oracle.input_pipeline import resized_crop_normalize
    from sparsify_clip_amd.input_pipeline import DeviceAugLoader, SyntheticCocoDataset
    ds = SyntheticCocoDataset(40, seed=3, pool=7)
    for train in (True, False):
        a = DeviceAugLoader(ds, 16, gpu, train=train, seed=5, workers=4)
        b = DeviceAugLoader(ds, 16, gpu, train=train, seed=5, workers=2)
        assert len(a) == 2                                           # drop_last: 40 // 16
        ba, bb = list(a), list(b)
        assert len(ba) == 2 and all(torch.equal(x[0], y[0]) and x[1] == y[1] for x, y in zip(ba, bb))
        assert ba[0][0].shape == (16, 3, 224, 224) and ba[0][0].dtype == torch.float32 and ba[0][0].is_cuda
        # replay the host-side draws of batch 0 and push them through the CPU pipeline
        c = DeviceAugLoader(ds, 16, gpu, train=train, seed=5)
        rng = np.random.Generator(np.random.Philox([5, 1000003]))
        order = rng.permutation(len(ds)) if train else np.arange(len(ds))
        r = c.batch_rng(0)
        for k, idx in enumerate(order[:16]):
            crop, flip, cap = c._geometry(*ds[int(idx)], r)
            want = resized_crop_normalize(np.ascontiguousarray(crop), (0, 0, crop.shape[0], crop.shape[1]), bool(flip), 224)
            assert torch.equal(ba[0][0][k].cpu(), want), (train, k)
            assert cap == ba[0][1][k] and cap in ds[int(idx)][1]
        if train:      # the next epoch draws new boxes
            assert not torch.equal(list(a)[0][0], ba[0][0])


@pytest.mark.timeout(120)
def test_loader_survives_an_abandoned_iteration(gpu):
    """A consumer that stops mid-epoch (break) leaves staging slots claimed by the producer thread; the loader hands them back and the
    next iteration runs to the end (it used to wait for a slot nobody would free)."""
    from sparsify_clip_amd.input_pipeline import DeviceAugLoader, SyntheticCocoDataset
    ds = SyntheticCocoDataset(64, seed=3, pool=7)
    loader = DeviceAugLoader(ds, 8, gpu, train=False, seed=5, workers=2)
    ref = [x[0].clone() for x in DeviceAugLoader(ds, 8, gpu, train=False, seed=5, workers=2)]
    assert len(ref) == 8
    for k, (images, _) in enumerate(loader):
        if k == 1:
            break
    again = [x[0].clone() for x in loader]
    assert len(again) == 8 and all(torch.equal(a, b) for a, b in zip(again, ref))


def test_coco_directory_through_the_runner_loader(gpu, tmp_path, monkeypatch):
    """A COCO-format fixture (PNG files + captions JSON in the reference's directory layout, :995-1001) read by CocoCaptionsDataset and
    served by train.dataset_loader with synthetic: False - the reference's dataset_loader(config) contract."""
    from PIL import Image
    from oracle.input_pipeline import resized_crop_normalize
    from sparsify_clip_amd.config import finalize_config
    from sparsify_clip_amd.train import dataset_loader
    rng = np.random.Generator(np.random.Philox(2))
    for split, n in (("train2017", 12), ("val2017", 8)):
        (tmp_path / "data/coco/images" / split).mkdir(parents=True)
        (tmp_path / "data/coco/annotations").mkdir(parents=True, exist_ok=True)
        images, anns = [], []
        for i in range(n):
            iid = 1000 - 7 * i                                      # descending ids: the dataset must sort them
            arr = rng.integers(0, 256, size=(60 + 5 * i, 90 - 3 * i, 3), dtype=np.uint8)
            Image.fromarray(arr, "RGB").save(tmp_path / "data/coco/images" / split / f"{iid:012d}.png")
            images.append({"id": iid, "file_name": f"{iid:012d}.png"})
            anns += [{"image_id": iid, "id": 10 * iid + c, "caption": f"caption {c} of image {iid}"} for c in range(5)]
        (tmp_path / "data/coco/annotations" / f"captions_{split}.json").write_text(json.dumps({"images": images, "annotations": anns}))
    monkeypatch.chdir(tmp_path)
    cfg = finalize_config({"project_name": "t", "run_name": "t", "seed": 42, "learning_rate": 1e-3, "batch_size": 4, "model": "ViT-B-32",
                           "num_train_samples": 8, "num_test_samples": -1, "epochs": 1, "loss_type": "anchor", "only_lunif_epochs": 0,
                           "anchor_temperature": 0.1, "anchor_temperature_learnable": False, "save_checkpoint_every_n_epochs": 20,
                           "resume_checkpoint": False, "fp16": True, "synthetic": False}, 0)
    train, test = dataset_loader(cfg, gpu)
    assert len(train) == 2 and len(test) == 2                        # Subset(range(8)) // 4 (:1033-1046); 8 val images // 4
    imgs, caps = next(iter(test))
    assert imgs.shape == (4, 3, 224, 224) and len(caps) == 4
    ids = sorted(1000 - 7 * i for i in range(8))
    for k in range(4):                                              # shuffle=False: ascending image ids, Resize((224,224))
        arr = np.asarray(Image.open(tmp_path / "data/coco/images/val2017" / f"{ids[k]:012d}.png").convert("RGB"))
        assert torch.equal(imgs[k].cpu(), resized_crop_normalize(arr, (0, 0, arr.shape[0], arr.shape[1]), False, 224))
        assert caps[k].endswith(f"of image {ids[k]}")
    timgs, tcaps = next(iter(train))
    assert timgs.shape == (4, 3, 224, 224) and all(isinstance(c, str) for c in tcaps) and torch.isfinite(timgs).all()
