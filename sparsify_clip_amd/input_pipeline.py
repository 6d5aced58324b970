"""Input pipeline of the training step: COCO captions reader, host-side batch assembly and DEVICE-side augmentation.

Reference: `dataset_loader` (sparsify_clip.py:992-1065) - torchvision `CocoCaptions` + transforms
(RandomResizedCrop((224,224)), RandomHorizontalFlip, ToTensor, Normalize; test: Resize((224,224)), ToTensor, Normalize),
a collate function that keeps ONE random caption of the (up to five) per image (:1050-1057), and 5 + 8 DataLoader worker
processes that hand fp32 [B,3,224,224] tensors to the training loop through pinned memory.

MI355X-first redesign of the same contract (iterable of (images fp32 [B,3,224,224] on the device, list[str] captions), drop_last,
`len()`):
  * the host only DECODES (Pillow) and draws the random crop boxes / flips; it ships the uint8 pixels of the crop boxes -
    about 1/8 of the bytes of the normalised fp32 batch - through one pinned staging buffer and one async H2D copy per batch;
  * crop + antialiased bilinear resize + flip + ToTensor + Normalize run on the GPU in two HIP kernels
    (`sc_image_resample_normalize`, csrc/augment.hip) that restate Pillow's resampler bit for bit, so the batch equals what
    the reference's CPU pipeline produces for the same boxes;
  * batch k+1 is assembled by worker threads and copied on a side stream while the step of batch k runs.
COCO itself is not available offline: `CocoCaptionsDataset` reads the standard directory layout when it exists
(./data/coco/..., as the reference), `SyntheticCocoDataset` produces uint8 images of COCO-like sizes with five captions each.
"""
from __future__ import annotations

import ctypes
import json
import math
import os
import queue
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from ._lib import LIB, ScError

MEAN = (0.48145466, 0.4578275, 0.40821073)   # reference :1003
STD = (0.26862954, 0.26130258, 0.27577711)   # reference :1004


# ------------------------------------------------------------------------------------------------ datasets
class CocoCaptionsDataset:
    """torchvision.datasets.CocoCaptions(root, annFile) without torchvision / pycocotools: item i = (uint8 [H,W,3] RGB array,
    list of caption strings) for the i-th image id in ascending order (as pycocotools-backed CocoDetection: sorted(imgs.keys()))."""

    def __init__(self, root: str, ann_file: str, limit: int = -1):
        with open(ann_file) as f:
            ann = json.load(f)
        self.root = root
        self.files = {im["id"]: im["file_name"] for im in ann["images"]}
        self.captions = {}
        for a in ann["annotations"]:
            self.captions.setdefault(a["image_id"], []).append(a["caption"])
        self.ids = sorted(self.files)
        if limit is not None and limit != -1:      # reference :1033-1046 Subset(range(n))
            self.ids = self.ids[:limit]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        from PIL import Image
        iid = self.ids[i]
        with Image.open(os.path.join(self.root, self.files[iid])) as im:
            arr = np.asarray(im.convert("RGB"))
        return arr, list(self.captions.get(iid, [""]))


class SyntheticCocoDataset:
    """Seeded stand-in with COCO's shape statistics: uint8 images (landscape 640x480 / portrait 480x640 / 640x427 ...) drawn
    from a small pool of distinct random images, five captions per image."""

    in_memory = True      # __getitem__ is a table lookup
    SIZES = [(480, 640), (640, 480), (427, 640), (640, 427), (375, 500), (500, 375)]
    WORDS = "a the man woman dog cat table street standing sitting on with of and red large small two people holding plate train".split()

    def __init__(self, n: int, seed: int = 0, pool: int = 48):
        rng = np.random.Generator(np.random.Philox(seed))
        self.n = n
        self.pool = [rng.integers(0, 256, size=(*self.SIZES[k % len(self.SIZES)], 3), dtype=np.uint8) for k in range(min(pool, max(n, 1)))]
        self.caps = [[" ".join(self.WORDS[j] for j in rng.integers(0, len(self.WORDS), size=int(rng.integers(5, 16)))) for _ in range(5)]
                     for _ in range(997)]      # a caption set per (index mod 997): cheap to serve, not periodic with the image pool

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.pool[i % len(self.pool)], self.caps[i % 997]


# ------------------------------------------------------------------------------------------------ host-side geometry
def random_resized_crop_params(rng: np.random.Generator, height: int, width: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision.transforms.RandomResizedCrop.get_params -> (top, left, h, w): ten attempts at a crop of `scale` x area with a
    log-uniform aspect ratio, then the centre-crop fallback."""
    area = height * width
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(lo, hi))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return int(rng.integers(0, height - h + 1)), int(rng.integers(0, width - w + 1)), h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


class _HostBatch:
    __slots__ = ("pixels", "nbytes", "offset", "dims", "flip", "tmp_offset", "tmp_bytes", "max_h", "captions", "n")


# ------------------------------------------------------------------------------------------------ loader
class DeviceAugLoader:
    """Iterable of (images fp32 [B,3,S,S] on `device`, captions list[str]) with the reference DataLoader's contract
    (`shuffle`, `drop_last=True`, `len()` = number of batches).  train=True: RandomResizedCrop + RandomHorizontalFlip; train=False:
    Resize.  Both keep one random caption per image (the reference passes the same collate_fn to both loaders, :1060-1061)."""

    def __init__(self, dataset, batch_size: int, device, train: bool, seed: int = 0, shuffle: bool | None = None, size: int = 224,
                 workers: int = 8, prefetch: int = 2, max_pixels_per_image: int = 640 * 480, tokenizer=None):
        self.ds, self.bs, self.device, self.train, self.size = dataset, int(batch_size), torch.device(device), bool(train), int(size)
        if self.device.type != "cuda":
            raise ScError("DeviceAugLoader needs a GPU device: the augmentation runs in libsparsify_hip.so")
        self.shuffle = train if shuffle is None else shuffle
        self.seed, self.epoch = seed, 0
        self.workers, self.prefetch = max(1, workers), max(1, prefetch)
        self.n_batches = len(dataset) // self.bs
        self.cap_bytes = self.bs * max_pixels_per_image * 3
        self.tokenizer = tokenizer      # optional: tokenise in the producer thread and yield int64 [B,ctx] device tensors instead of strings
        self._copy_stream = None
        self._stage = None

    def __len__(self):
        return self.n_batches

    # ---- host side
    def _geometry(self, img, caps, rng: np.random.Generator):
        """The random draws of one sample, in the reference's order: crop box (:1009), flip (:1010), caption choice (:1055)."""
        h, w = img.shape[0], img.shape[1]
        if self.train:
            top, left, bh, bw = random_resized_crop_params(rng, h, w)
            flip = int(rng.random() < 0.5)
        else:
            top, left, bh, bw, flip = 0, 0, h, w, 0                         # :1014 Resize((224,224))
        if bh > 31 * self.size or bw > 31 * self.size:
            raise ScError(f"a {bh}x{bw} box is more than a 31x down-scale to {self.size} (kernel tap limit)")
        cap = caps[int(rng.integers(0, len(caps)))]
        return img[top:top + bh, left:left + bw], flip, cap

    def batch_rng(self, batch_index: int) -> np.random.Generator:
        """One Philox stream per (seed, epoch, batch): samples consume it in batch order, whatever the number of worker threads."""
        return np.random.Generator(np.random.Philox([self.seed, self.epoch * (1 << 32) + batch_index]))

    def _assemble(self, batch_index: int, order, pool: ThreadPoolExecutor, pinned: torch.Tensor) -> _HostBatch:
        ids = order[batch_index * self.bs:(batch_index + 1) * self.bs]
        if getattr(self.ds, "in_memory", False):                            # nothing to decode: 1024 pool tasks would only pass the GIL around
            items = [self.ds[int(i)] for i in ids]
        else:
            items = list(pool.map(lambda i: self.ds[int(i)], ids))          # decode in parallel (Pillow releases the GIL)
        rng = self.batch_rng(batch_index)
        samples = [self._geometry(img, caps, rng) for img, caps in items]   # cheap scalar draws, sequential = deterministic
        hb = _HostBatch()
        hb.n = len(samples)
        sizes = np.array([s[0].shape[0] * s[0].shape[1] * 3 for s in samples], dtype=np.int64)
        hb.offset = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        hb.nbytes = int(sizes.sum())
        if hb.nbytes > pinned.numel():
            raise ScError(f"batch needs {hb.nbytes} staging bytes, the loader reserved {pinned.numel()} (max_pixels_per_image)")
        hb.dims = np.array([[s[0].shape[0], s[0].shape[1], 0, 0, s[0].shape[0], s[0].shape[1]] for s in samples], dtype=np.int32)
        hb.flip = np.array([s[1] for s in samples], dtype=np.int32)
        tmp_sizes = hb.dims[:, 4].astype(np.int64) * self.size * 3
        hb.tmp_offset = np.concatenate([[0], np.cumsum(tmp_sizes)[:-1]]).astype(np.int64)
        hb.tmp_bytes = int(tmp_sizes.sum())
        hb.max_h = int(hb.dims[:, 4].max())
        hb.captions = [s[2] for s in samples]
        if self.tokenizer is not None:
            hb.captions = self.tokenizer(hb.captions).pin_memory()
        # strided crop views -> contiguous staging bytes: one native call (sc_host_gather_rows, C++ threads, GIL released by ctypes)
        crops = [s[0] if s[0].strides[1:] == (3, 1) else np.ascontiguousarray(s[0]) for s in samples]
        n = hb.n
        src = (ctypes.c_void_p * n)(*[a.ctypes.data for a in crops])
        stride = np.array([a.strides[0] for a in crops], dtype=np.int64)
        rows = np.ascontiguousarray(hb.dims[:, 4], dtype=np.int64)
        row_bytes = hb.dims[:, 5].astype(np.int64) * 3
        LIB.call("sc_host_gather_rows", n, src, stride.ctypes.data, rows.ctypes.data, row_bytes.ctypes.data, pinned.data_ptr(),
                 hb.offset.ctypes.data, pinned.numel(), max(1, self.workers // 2))      # half the threads: the copy is memory-bound, the
        #                                                                                       training thread needs a core to enqueue the step
        hb.pixels = pinned
        return hb

    # ---- device side
    def __iter__(self):
        if self.n_batches == 0:
            return
        rng = np.random.Generator(np.random.Philox([self.seed, 1000003 + self.epoch]))
        order = rng.permutation(len(self.ds)) if self.shuffle else np.arange(len(self.ds))
        dev = self.device
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
            nslots = self.prefetch + 1
            self._stage = [dict(pinned=torch.empty(self.cap_bytes, dtype=torch.uint8).pin_memory(),
                                pixels=torch.empty(self.cap_bytes, dtype=torch.uint8, device=dev),
                                meta=torch.empty(self.bs * 48, dtype=torch.uint8, device=dev),
                                meta_pinned=torch.empty(self.bs * 48, dtype=torch.uint8).pin_memory(),
                                copied=torch.cuda.Event(), consumed=torch.cuda.Event(), free=threading.Event()) for _ in range(nslots)]
            for s in self._stage:
                s["free"].set()
        q: queue.Queue = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def producer():
            try:
                with ThreadPoolExecutor(max_workers=self.workers) as pool:
                    for b in range(self.n_batches):
                        slot = self._stage[b % len(self._stage)]
                        while not slot["free"].wait(0.1):
                            if stop.is_set():
                                return
                        slot["free"].clear()
                        q.put((b, self._assemble(b, order, pool, slot["pinned"])))
                q.put(None)
            except BaseException as e:      # surfaces in the consumer
                q.put(e)

        # The assembly thread runs ~10-25 ms of pure Python per batch.  Every kernel launch of the training thread drops the GIL (ctypes,
        # torch) and, with CPython's default 5 ms switch interval, waits up to 5 ms to get it back while the assembly thread is in Python:
        # measured 42 ms instead of 5 ms to enqueue a step (tools/pipeline_bench.py).  A 50 us interval bounds each hand-back instead.
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 5e-5))
        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            pending = None      # (slot, host batch) whose H2D copy is in flight

            def start_copy(item):
                b, hb = item
                slot = self._stage[b % len(self._stage)]
                with torch.cuda.stream(self._copy_stream):
                    self._copy_stream.wait_event(slot["consumed"])          # the kernels of the batch that used this slot are done
                    slot["pixels"][: hb.nbytes].copy_(hb.pixels[: hb.nbytes], non_blocking=True)
                    meta = np.concatenate([hb.offset.view(np.uint8), hb.tmp_offset.view(np.uint8), hb.dims.reshape(-1).view(np.uint8),
                                           hb.flip.view(np.uint8)])
                    slot["meta_pinned"][: meta.size].copy_(torch.from_numpy(meta))
                    slot["meta"][: meta.size].copy_(slot["meta_pinned"][: meta.size], non_blocking=True)
                    slot["copied"].record(self._copy_stream)
                return slot, hb

            def get():
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                return item

            item = get()
            if item is not None:
                pending = start_copy(item)
            while pending is not None:
                slot, hb = pending
                nxt = get()
                pending = start_copy(nxt) if nxt is not None else None       # batch k+1 travels while batch k is augmented and trained on
                cur = torch.cuda.current_stream()
                cur.wait_event(slot["copied"])
                n = hb.n
                m = slot["meta"]
                off = m[: n * 8].view(torch.int64)
                toff = m[n * 8: n * 16].view(torch.int64)
                dims = m[n * 16: n * 16 + n * 24].view(torch.int32)
                flip = m[n * 40: n * 40 + n * 4].view(torch.int32)
                images = ops.image_resample_normalize(slot["pixels"], off, dims, flip, toff, n, hb.max_h, hb.tmp_bytes, self.size, MEAN, STD)
                slot["consumed"].record(cur)
                # the pinned buffer may be refilled as soon as the copy has been issued AND completed: free it once the copy event is done
                slot["copied"].synchronize()
                slot["free"].set()
                caps = hb.captions.to(dev, non_blocking=True) if torch.is_tensor(hb.captions) else hb.captions
                yield images, caps
            self.epoch += 1
        finally:
            sys.setswitchinterval(old_interval)
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(0.05)
            # an abandoned iteration (break, exception, closed generator) leaves slots the producer had claimed and copies that may still
            # be in flight: wait for the copy stream and hand every slot back, so that the next __iter__ starts from a clean stage
            # instead of waiting for a slot nobody will free
            self._copy_stream.synchronize()
            for s_ in self._stage:
                s_["free"].set()


def coco_paths(split: str):
    """The reference's hard-coded layout (:995-1001)."""
    name = "train2017" if split == "train" else "val2017"
    return f"./data/coco/images/{name}/", f"./data/coco/annotations/captions_{name}.json"
