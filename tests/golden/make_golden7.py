#!/usr/bin/env python3
"""Seventh golden generator — gradients of the DNPH step loss: the REFERENCE's DNPH_out (train/DNPH_TOMM/loss.py) plus the
`- 0.1 * noise_loss` term exactly as train/DNPH_TOMM/hash_train.py:65-81 forms it, differentiated by torch autograd on the CPU
w.r.t. both hash outputs, both classifier outputs and the proxies.  Inputs are the dnph cases of make_golden2.py (heads2util.py)
and the noise rows already recorded in dnph.npz."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import heads2util as hu  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402


def gen():
    loss_mod = ref_import("train.DNPH_TOMM.loss")
    g = np.load(os.path.join(HERE, "dnph.npz"))
    out = {}
    for (B, K, C) in hu.DNPH_CASES:
        c = hu.dnph_case(B, K, C)
        tag = c["tag"]
        m = loss_mod.DNPH_out(SimpleNamespace(nclass=C, output_dim=K))
        m.proxies.data.copy_(t(c["prox"]))
        hi, ht, pi, pt = (t(c[k]).requires_grad_() for k in ("hi", "ht", "pi", "pt"))
        lab = t(c["lab"])
        ni, nt = t(g[f"{tag}_noise_i"].astype(np.float32)), t(g[f"{tag}_noise_t"].astype(np.float32))
        noise_loss = hi.mul(ni).sum(dim=-1).mean() + ht.mul(nt).sum(dim=-1).mean()
        loss = m(hi, ht, pi, pt, lab, lab) - 0.1 * noise_loss
        loss.backward()
        out[f"{tag}_step_loss"] = loss.detach().numpy()
        for name, v in (("ghi", hi), ("ght", ht), ("gpi", pi), ("gpt", pt), ("gprox", m.proxies)):
            out[f"{tag}_{name}"] = v.grad.numpy()
    save("dnph_grads.npz", **out)


if __name__ == "__main__":
    import tempfile
    install_stubs()
    os.chdir(tempfile.mkdtemp())
    torch.manual_seed(0)
    gen()
