"""Shared by tests/golden/make_golden6.py and the GPU test: the seeded problem of the DCHMT training-trajectory golden
(configs[0] in miniature: tiny CLIP, select hash heads, 16 bits, batch 8, euclidean / l2, BertAdam as the trainer builds it)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import recipe  # noqa: E402

CFG, SEED, B, L, K, C, STEPS = recipe.CLIP_TINY, 7, 8, 16, 16, 10, 4
OPT = dict(lr=1e-3, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=8, weight_decay=0.2, max_grad_norm=1.0)
CLIP_LR = 1e-5
LOSS = dict(vartheta=0.5, sim_threshold=0.1, similarity_function="euclidean", loss_type="l2")


def fill_head(head, seed):
    """deterministic weights for a HashLayer (fc + K two-way Linears), identical for the reference's and the mirror's class"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(head.named_parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.01))


def batch(step):
    img = torch.from_numpy(recipe.images(B, CFG["image_resolution"], SEED + 10 * step))
    txt = torch.from_numpy(recipe.captions(B, L, CFG["vocab_size"], SEED + 10 * step))
    lab = torch.from_numpy(recipe.labels(B, C, SEED + step, p=0.25, tag=f"dchmt_traj_{step}"))
    return img, txt, lab


def cut(a):
    a = np.asarray(a).reshape(-1)
    return a[::5].copy() if a.size > 2000 else a.copy()
