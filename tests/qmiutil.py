"""Seeded DNpH (qmi_loss) cases shared by tests/golden/make_golden15.py and the tests."""
import numpy as np

import recipe

CASES = [(8, 16, 24, 0.15), (48, 32, 80, 0.05), (256, 64, 24, 0.15), (16, 128, 21, 0.0)]


def qmi_case(B, K, C, p, seed=91):
    """p = 0: one-hot labels, no two samples share a class (only the diagonal of the indicator is set)"""
    tag = f"B{B}_K{K}_C{C}"
    lab = recipe.labels(B, C, seed, p=p, tag=f"qmi_lab_{tag}") if p > 0 else np.eye(B, C, dtype=np.float32)
    return dict(tag=tag, x=np.tanh(recipe.features(B, K, seed, f"qmi_x_{tag}")), y=np.tanh(recipe.features(B, K, seed, f"qmi_y_{tag}")),
                lab=lab)
