#!/usr/bin/env python3
"""Sixteenth golden generator — DMsH-LN: the REFERENCE's LabelNet (train/DMsH_LN/labelnet.py:6-21) and MultiSimilarityLoss
(train/DMsH_LN/MSLOSS.py:4-55) with their autograd gradients on the CPU, called the way the trainer calls them
(train/DMsH_LN/hash_train.py:53-63): label codes from LabelNet, then MSL(img, codes), MSL(txt, codes), MSL(img, codes, feat2=txt).
Cases: miniature, 48 x 32 x 80, the bench batch (256 x 64 x 24), and a batch in which every row is skipped (no negative pair)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import recipe  # noqa: E402,F401
from make_golden import install_stubs, ref_import, save, t  # noqa: E402

from mslutil import CASES, msl_case  # noqa: E402


def gen():
    msl_mod = ref_import("train.DMsH_LN.MSLOSS")
    ln_mod = ref_import("train.DMsH_LN.labelnet")
    out = {}
    for (B, K, C, p, epoch) in CASES:
        c = msl_case(B, K, C, p, epoch)
        tag = c["tag"]
        net = ln_mod.LabelNet(label_dim=C, code_len=K)
        with torch.no_grad():
            net.fc1.weight.copy_(t(c["w1"])); net.fc1.bias.copy_(t(c["b1"]))
            net.fc2.weight.copy_(t(c["w2"])); net.fc2.bias.copy_(t(c["b2"]))
        net.set_alpha(epoch)
        feat, hid, code = net(t(c["lab"]), device="cpu")
        out[f"{tag}_ln_feat"], out[f"{tag}_ln_hid"], out[f"{tag}_ln_code"] = feat.detach().numpy(), hid.detach().numpy(), code.detach().numpy()
        msl = msl_mod.MultiSimilarityLoss()
        x, y = t(c["x"]).requires_grad_(), t(c["y"]).requires_grad_()
        l_ii = msl(x, code.detach())
        l_tt = msl(y, code.detach())
        l_it = msl(x, code.detach(), feat2=y)
        for name, l in (("ii", l_ii), ("tt", l_tt), ("it", l_it)):
            out[f"{tag}_loss_{name}"] = l.detach().numpy()
        total = l_ii + l_tt + l_it
        if total.requires_grad and total.grad_fn is not None:
            total.backward()
        out[f"{tag}_gx"] = x.grad.numpy() if x.grad is not None else np.zeros_like(c["x"])
        out[f"{tag}_gy"] = y.grad.numpy() if y.grad is not None else np.zeros_like(c["y"])
        print(tag, float(l_ii), float(l_tt), float(l_it), float(np.abs(out[f"{tag}_gx"]).max()))
    save("msl.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
