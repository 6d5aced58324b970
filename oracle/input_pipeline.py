"""TEST INFRASTRUCTURE ONLY - CPU oracle of the input stage (reference sparsify_clip.py:992-1065).

The reference builds its batches with torchvision transforms on PIL images.  torchvision is absent here, Pillow (the library
those transforms call for PIL inputs) is present, so the oracle restates torchvision's functional code on top of the REAL
Pillow resampler:
    RandomResizedCrop((S,S))  = F.resized_crop(img, top, left, h, w, (S,S), BILINEAR) = img.crop(...).resize((S,S), BILINEAR)
    Resize((S,S))             = img.resize((S,S), BILINEAR)
    RandomHorizontalFlip      = img.transpose(FLIP_LEFT_RIGHT) with probability 0.5
    ToTensor                  = uint8 HWC -> float32 CHW / 255
    Normalize(mean, std)      = (x - mean) / std
The crop-box sampler follows torchvision.transforms.RandomResizedCrop.get_params (scale (0.08, 1), log-uniform ratio (3/4, 4/3),
ten attempts, centre-crop fallback); its random stream is torch's in the reference and a numpy Philox stream here (parity of the
stream itself is unpinned - torchvision cannot be imported - the geometry and pixel arithmetic are pinned by Pillow)."""
import math

import numpy as np
import torch
from PIL import Image

MEAN = (0.48145466, 0.4578275, 0.40821073)   # reference :1003-1004
STD = (0.26862954, 0.26130258, 0.27577711)


def resized_crop_normalize(img_u8: np.ndarray, box, flip: bool, size: int = 224) -> torch.Tensor:
    """img_u8 [H,W,3] uint8, box = (top, left, h, w) -> float32 [3,size,size], exactly the reference's train transform."""
    top, left, h, w = box
    im = Image.fromarray(img_u8, "RGB").crop((left, top, left + w, top + h)).resize((size, size), Image.BILINEAR)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    x = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(STD, dtype=torch.float32)[:, None, None]
    return x.sub_(mean).div_(std)


def random_resized_crop_params(rng: np.random.Generator, height: int, width: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params -> (top, left, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            top = int(rng.integers(0, height - h + 1))
            left = int(rng.integers(0, width - w + 1))
            return top, left, h, w
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
