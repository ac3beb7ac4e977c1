#!/usr/bin/env python3
"""Fifteenth golden generator — DNpH: the REFERENCE's qmi_loss (train/DNpH_TMM/loss.py:5-72) and its autograd gradients on the
CPU.  The reference casts the indicator with `torch.cuda.FloatTensor` (:41), which does not exist in a CPU-only process: the
generator points that name at torch.FloatTensor for the duration of the call (an environment stand-in; no reference code is
altered).  Cases: miniature, the bench batch (256 x 64 x 24), a batch without any label-sharing pair off the diagonal."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import recipe  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402

from qmiutil import CASES, qmi_case  # noqa: E402


def gen():
    loss_mod = ref_import("train.DNpH_TMM.loss")
    torch.cuda.FloatTensor = torch.FloatTensor
    out = {}
    for (B, K, C, p) in CASES:
        c = qmi_case(B, K, C, p)
        x, y = t(c["x"]).requires_grad_(), t(c["y"]).requires_grad_()
        loss = loss_mod.qmi_loss(images=x, texts=y, targets=t(c["lab"]))
        loss.backward()
        tag = c["tag"]
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_gx"] = x.grad.numpy()
        out[f"{tag}_gy"] = y.grad.numpy()
    save("qmi.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
