#!/usr/bin/env python3
"""Seventeenth golden generator — DHaPH: the REFERENCE's self-paced contrastive loss (train/DHaPH/MSLoss.py:6-33) with its autograd
gradients on the CPU, called the way the trainer calls it (train/DHaPH/hash_train.py:68-70): MSLoss(img, img), MSLoss(txt, txt),
MSLoss(img, txt) at `epoch + 1`."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import recipe  # noqa: E402,F401
from make_golden import install_stubs, ref_import, save, t  # noqa: E402

from mslutil import SPL_CASES, spl_case  # noqa: E402


def gen():
    mod = ref_import("train.DHaPH.MSLoss")
    out = {}
    for (B, K, C, p, epoch, total) in SPL_CASES:
        c = spl_case(B, K, C, p, epoch, total)
        tag = c["tag"]
        crit = mod.MSLoss(temperature=0.3, totalepoch=total, self_paced=True)
        x, y = t(c["x"]).requires_grad_(), t(c["y"]).requires_grad_()
        lab = t(c["lab"])
        l1, l2, l3 = crit(x, x, lab, epoch), crit(y, y, lab, epoch), crit(x, y, lab, epoch)
        (l1 + l2 + l3).backward()
        for name, l in (("ii", l1), ("tt", l2), ("it", l3)):
            out[f"{tag}_loss_{name}"] = l.detach().numpy()
        out[f"{tag}_gx"], out[f"{tag}_gy"] = x.grad.numpy(), y.grad.numpy()
        plain = mod.MSLoss(temperature=0.3, totalepoch=total, self_paced=False)
        out[f"{tag}_loss_it_plain"] = plain(t(c["x"]), t(c["y"]), lab, epoch).numpy()
        print(tag, float(l1), float(l2), float(l3), float(np.abs(out[f"{tag}_gx"]).max()))
    save("spl.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
