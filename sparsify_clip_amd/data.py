"""Input side of the training step: synthetic (image, caption) batches and a dependency-free tokenizer.

The reference streams COCO captions through torchvision + 13 DataLoader workers and tokenises with open_clip's
BPE (sparsify_clip.py:992-1065, :692, :762).  Neither COCO, torchvision nor the BPE vocabulary file exists offline,
so this build feeds the step from a seeded synthetic source with the same contract:
images fp32 [B,3,224,224] at post-Normalize scale, captions as int64 [B,77] token rows
[SOT, w_1..w_L, EOT, 0...] (EOT = 49407 is the unique maximum, so argmax pooling is well defined).
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

SOT, EOT, CTX, VOCAB = 49406, 49407, 77, 49408


class HashTokenizer:
    """callable(list[str]) -> LongTensor[B, ctx].  NOT open_clip's BPE (its vocabulary file is absent offline):
    lower-cased whitespace words are hashed (crc32) into [1, SOT) - stable across runs and processes.  Same framing as
    open_clip's SimpleTokenizer: SOT first, EOT after the last word, zero padding, truncation keeps EOT."""

    def __init__(self, context_length=CTX, vocab=VOCAB):
        self.context_length, self.vocab = context_length, vocab
        self.sot, self.eot = vocab - 2, vocab - 1
        self._ids = {}      # word -> id (the hash is a pure function of the word; the table only saves recomputing it)

    def __call__(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        ctx, ids_of, mod = self.context_length, self._ids, self.sot - 1
        if len(ids_of) > (1 << 20):
            ids_of.clear()
        out = np.zeros((len(texts), ctx), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot]
            for w in t.lower().split():
                v = ids_of.get(w)
                if v is None:
                    v = ids_of[w] = 1 + zlib.crc32(w.encode("utf-8")) % mod
                ids.append(v)
            ids.append(self.eot)
            if len(ids) > ctx:
                ids = ids[:ctx]
                ids[-1] = self.eot
            out[i, : len(ids)] = ids
        return torch.from_numpy(out)


def caption_length(tokens) -> int:
    """Longest caption of a tokenised batch, EOT included (EOT is the largest id: open_clip pools at argmax, reference :769)."""
    return int(tokens.argmax(dim=1).max()) + 1


def get_tokenizer(model_name=None, context_length=CTX, vocab=VOCAB):
    """Mirror of open_clip.get_tokenizer(name) (reference :692, :560)."""
    return HashTokenizer(context_length, vocab)


def synthetic_batch(seed: int, batch: int, image_size=224, ctx=CTX, vocab=VOCAB):
    """numpy Philox stream -> (images fp32 [B,3,R,R] ~ N(0,1), tokens int64 [B,ctx]); lengths L ~ U{5..30}."""
    rng = np.random.Generator(np.random.Philox(seed))
    images = rng.standard_normal((batch, 3, image_size, image_size), dtype=np.float32)
    sot, eot = vocab - 2, vocab - 1
    tokens = np.zeros((batch, ctx), dtype=np.int64)
    hi = min(30, ctx - 3)
    lens = rng.integers(min(5, hi), hi + 1, size=batch)
    for i, n in enumerate(lens):
        tokens[i, 0] = sot
        tokens[i, 1:1 + n] = rng.integers(1, sot, size=n)
        tokens[i, 1 + n] = eot
    return torch.from_numpy(images), torch.from_numpy(tokens)


class SyntheticLoader:
    """Finite iterable of pre-generated device batches with `len()`, standing in for the reference's DataLoader
    (drop_last=True semantics: exactly num_samples // batch_size batches).  Yields (images, tokens); a batch whose second
    element is a list[str] would be tokenised by the train loop exactly as the reference does (:762)."""

    def __init__(self, num_samples, batch_size, seed, device, image_size=224, ctx=CTX, vocab=VOCAB, distinct=2):
        self.n_batches = num_samples // batch_size
        self.device = device
        self.pool = [tuple(t.to(device) for t in synthetic_batch(seed + k, batch_size, image_size, ctx, vocab)) for k in range(min(distinct, max(self.n_batches, 1)))]

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for i in range(self.n_batches):
            yield self.pool[i % len(self.pool)]
