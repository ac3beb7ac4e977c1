"""Synthetic stand-in for the reference's data layer (dataset/base.py, dataset/dataloader.py — host
input pipeline, out of scope).  Honours the path's INPUT CONTRACT:
  __getitem__ -> (image f32 [3,R,R] ~N(0,1), caption i64 [maxWords] SOT..EOT,0-pad, label f32 [C], index)
  get_all_label() -> [n, C]
and the reference's split rule (dataset/dataloader.py:6-11): seeded permutation, first query_num =
queries, next train_num = train, everything after the queries (train included) = retrieval DB."""
import numpy as np
import torch
from torch.utils.data import Dataset

SOT, EOT = 49406, 49407


class SyntheticPairs(Dataset):
    """signal = 0: image and caption are independent noise (throughput / contract tests).  signal > 0: every class owns a fixed
    image pattern and a few caption tokens, an item shows the patterns / tokens of its labels plus noise — something a model can
    learn, so a few epochs must raise the mAP (tests/test_gpu_heads_losses.py::test_training_learns)."""
    signal = 0.0
    # tile > 0: a class's image pattern is ONE tile x tile block repeated over the image (tile = the ViT's patch size: every patch
    # embedding then moves the same way, which survives the near-uniform attention of a random-init tower - a full-size noise
    # pattern averages out over the 49 patches and every image collapses onto one hash code); 0: a full-size pattern
    tile = 0
    # caption_tokens > 0 (with signal > 0): a caption is MADE of its classes' tokens - every class owns this many, the body cycles
    # through the tokens of the item's classes with one position in five left to noise.  Three class tokens dropped into ~15 random
    # ones (the default) vanish in the causal averaging of a random-init text tower: round 4's configs[0] database held 13 distinct
    # text codes in 1 000 items
    caption_tokens = 0

    def __init__(self, ids, labels, max_words, resolution, seed):
        self.ids, self.labels = ids, labels
        self.max_words, self.resolution, self.seed = max_words, resolution, seed

    def __len__(self):
        return len(self.ids)

    def get_all_label(self):
        return torch.from_numpy(self.labels)

    def __getitem__(self, index):
        rng = np.random.default_rng([self.seed, int(self.ids[index])])
        image = rng.standard_normal((3, self.resolution, self.resolution)).astype(np.float32)
        cap = np.zeros(self.max_words, np.int64)
        n = int(rng.integers(2, self.max_words))
        cap[0] = SOT
        cap[1:n] = rng.integers(1, SOT, size=n - 1)
        cap[n] = EOT
        if self.signal > 0:
            classes = np.nonzero(self.labels[index])[0]
            for c in classes:
                proto = np.random.default_rng([self.seed, 7919, int(c)])
                if self.tile > 0:
                    reps = -(-self.resolution // self.tile)
                    block = proto.standard_normal((3, self.tile, self.tile)).astype(np.float32)
                    image += self.signal * np.tile(block, (1, reps, reps))[:, :self.resolution, :self.resolution]
                else:
                    image += self.signal * proto.standard_normal(image.shape).astype(np.float32)
                words = proto.integers(1, SOT, size=3)                       # the class's caption tokens
                if self.caption_tokens == 0:
                    for w in words:
                        cap[int(rng.integers(1, max(n, 2)))] = w
            if self.caption_tokens > 0 and len(classes) and n > 1:
                own = np.concatenate([np.random.default_rng([self.seed, 104729, int(c)]).integers(1, SOT, size=self.caption_tokens)
                                      for c in classes])
                body = own[np.arange(n - 1) % len(own)]
                noise = rng.random(n - 1) < 0.2
                body[noise] = cap[1:n][noise]
                cap[1:n] = body
        return torch.from_numpy(image), torch.from_numpy(cap), torch.from_numpy(self.labels[index]), index


def dataloader(total, nclass, maxWords=32, imageResolution=224, query_num=5000, train_num=10000, seed=None):
    rng = np.random.default_rng(seed)
    labels = (rng.random((total, nclass)) < 0.15).astype(np.float32)
    perm = np.random.RandomState(seed).permutation(total)
    q, t, r = perm[:query_num], perm[query_num:query_num + train_num], perm[query_num:]
    mk = lambda ids: SyntheticPairs(ids, labels[ids], maxWords, imageResolution, seed or 0)
    return mk(t), mk(q), mk(r)


class SyntheticPairsMITH(SyntheticPairs):
    """train/MITH/data.py contract: (image, caption, key_padding_mask = caption == 0, label, index)."""

    def __getitem__(self, index):
        image, cap, label, index = super().__getitem__(index)
        return image, cap, cap == 0, label, index


def dataloader_mith(total, nclass, maxWords=32, imageResolution=224, query_num=5000, train_num=10000, seed=None):
    tr, q, r = dataloader(total, nclass, maxWords, imageResolution, query_num, train_num, seed)
    for d in (tr, q, r):
        d.__class__ = SyntheticPairsMITH
    return tr, q, r
