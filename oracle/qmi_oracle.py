"""ORACLE (test infrastructure only) — DNpH's qmi_loss, reference train/DNpH_TMM/loss.py:5-72 in its default configuration
(use_cosine=True, use_square_clamp=True, M = B^2 / sum(D)), restated in numpy float32 with the reference's operation order.
Pinned by tests/golden/qmi.npz (the reference's own loss values and autograd gradients, tests/golden/make_golden15.py)."""
import numpy as np


def qmi_loss(images, texts, targets, eps=1e-8):
    f = np.float32
    images, texts, targets = (np.asarray(a, dtype=f) for a in (images, texts, targets))
    images = images / (np.sqrt(np.sum(images ** 2, axis=1, keepdims=True)) + f(eps))          # :18
    Y = f(0.5) * (images @ images.T + f(1))                                                    # :19-20
    texts = texts / (np.sqrt(np.sum(texts ** 2, axis=1, keepdims=True)) + f(eps))              # :22
    T = f(0.5) * (texts @ texts.T + f(1))                                                      # :23-24
    YT = f(0.5) * (images @ texts.T + f(1))                                                    # :26-27
    D = (targets @ targets.T > 0).astype(f)                                                    # :40-41
    with np.errstate(divide="ignore"):
        M = f(D.shape[1] ** 2) / np.sum(D, dtype=f)                                            # :43-44  (inf when no pair shares a label)
    inv_m = f(1.0) / M
    total = f(0)
    for S in (Y, T, YT):                                                                       # :46-57
        total = total + np.sum((D * S - f(1)) ** 2 + inv_m * S ** 2, dtype=f)
    return f(total)
