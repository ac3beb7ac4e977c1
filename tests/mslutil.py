"""Seeded DMsH-LN cases (multi-similarity loss + LabelNet) shared by tests/golden/make_golden16.py and the tests."""
import numpy as np

import recipe

CASES = [(8, 16, 24, 0.3, 0), (48, 32, 80, 0.05, 3), (256, 64, 24, 0.15, 7), (6, 16, 4, 1.0, 1)]     # (B, K, C, p, epoch)


def msl_case(B, K, C, p, epoch, seed=131):
    """p = 1: every sample carries every class -> LabelNet gives one code for all, no negative pair, every row is skipped"""
    tag = f"B{B}_K{K}_C{C}"
    hid = (C + K) // 2
    r = lambda shape, name, s: (s * recipe._rng(seed, f"msl_{name}_{tag}").standard_normal(shape)).astype(np.float32)
    lab = recipe.labels(B, C, seed, p=p, tag=f"msl_lab_{tag}") if p < 1 else np.ones((B, C), np.float32)
    w1, b1, w2 = r((hid, C), "w1", 1.0), r((hid,), "b1", 0.5), r((K, hid), "w2", hid ** -0.5)
    # fc2's bias removes the batch mean of its input: without it the ReLU features' common positive part makes every pair of
    # label codes "similar" (code_i . code_j > 0) and the loss skips every row
    b2 = (-(np.maximum(lab @ w1.T + b1, 0).mean(0) @ w2.T)).astype(np.float32)
    return dict(tag=tag, epoch=epoch, lab=lab, w1=w1, b1=b1, w2=w2, b2=b2,
                x=np.tanh(recipe.features(B, K, seed, f"msl_x_{tag}")), y=np.tanh(recipe.features(B, K, seed, f"msl_y_{tag}")))


# DHaPH self-paced contrastive loss (train/DHaPH/MSLoss.py): (B, K, C, p, epoch, total_epochs)
SPL_CASES = [(8, 16, 24, 0.3, 1, 100), (48, 32, 80, 0.08, 20, 100), (256, 64, 24, 0.15, 50, 100), (32, 128, 21, 0.2, 3, 2)]


def spl_case(B, K, C, p, epoch, total, seed=151):
    tag = f"B{B}_K{K}_C{C}_e{epoch}"
    lab = recipe.labels(B, C, seed, p=p, tag=f"spl_lab_{tag}")
    lab[lab.sum(1) == 0, 0] = 1.0            # an all-zero label row has no positive at all, not even itself: log(0) upstream
    return dict(tag=tag, lab=lab, x=np.tanh(recipe.features(B, K, seed, f"spl_x_{tag}")), y=np.tanh(recipe.features(B, K, seed, f"spl_y_{tag}")))
