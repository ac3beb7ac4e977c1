#!/usr/bin/env python3
"""Fifth golden generator — SURVEY §8f "next" #2: parameter gradients of the REFERENCE's CLIP towers (model/base/model.py,
torch autograd on the CPU, fp32) for L = sum(encode_image(image) * Gi) + sum(encode_text(text) * Gt) on the tiny configuration.
Every gradient tensor is recorded (tensors above 4096 elements every 7th element)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402
from make_golden import build_ref_clip, install_stubs, save, t  # noqa: E402

SLICE = 7


def cut(a):
    a = np.asarray(a).reshape(-1)
    return a[::SLICE].copy() if a.size > 4096 else a.copy()


def cotangents(B, E, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, E, generator=g), torch.randn(B, E, generator=g)


def gen():
    cfg, seed, B, L = recipe.CLIP_TINY, 7, 3, 16
    clip = build_ref_clip(cfg, seed)
    image = t(recipe.images(B, cfg["image_resolution"], seed))
    text = t(recipe.captions(B, L, cfg["vocab_size"], seed))
    gi, gt = cotangents(B, cfg["embed_dim"], 23)
    fi = clip.encode_image(image)
    ft = clip.encode_text(text)
    ((fi * gi).sum() + (ft * gt).sum()).backward()
    out = {"img_feat": fi.detach().numpy(), "txt_feat": ft.detach().numpy()}
    names = []
    for name, p in clip.named_parameters():
        if p.grad is None:
            continue                      # logit_scale
        names.append(name)
        out["g_" + name] = cut(p.grad.numpy())
        out["n_" + name] = np.float64(p.grad.double().norm().item())
    out["names"] = np.array(names)
    save("clip_tiny_grads.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
