#!/usr/bin/env python3
"""Eleventh golden generator — parameter gradients of the REFERENCE's MITH trunk (model/MITH.py: CLIP1 through its own
build_model, every token projected, key_padding_mask in the text blocks) by torch autograd on the CPU for
L = sum(seq_i * G1) + sum(cls_i * G2) + sum(seq_t * G3) + sum(eos_t * G4) on the tiny 512-embed configuration (inputs of
make_golden3.py).  Every gradient tensor is recorded (tensors above 4096 elements every 7th element) with its norm."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import recipe  # noqa: E402
import mithutil as mu  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402
from make_golden5 import cut  # noqa: E402


def gen():
    mm = ref_import("model.MITH")
    cfg, seed = mu.CLIP_TINY512, 7
    sd = {k: t(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}
    clip = mm.build_model(sd).float()
    image = t(recipe.images(3, cfg["image_resolution"], seed))
    text = recipe.captions(3, 16, cfg["vocab_size"], seed)
    kpm = (text == 0)
    seq_i, _, cls_i = clip.encode_image(image)
    seq_t, _, new_kpm, eos_t = clip.encode_text(t(text), t(kpm))
    g = torch.Generator().manual_seed(31)
    G = [torch.randn(x.shape, generator=g) for x in (seq_i, cls_i, seq_t, eos_t)]
    ((seq_i * G[0]).sum() + (cls_i * G[1]).sum() + (seq_t * G[2]).sum() + (eos_t * G[3]).sum()).backward()
    out = {"seq_i": seq_i.detach().numpy(), "cls_i": cls_i.detach().numpy(), "eos_t": eos_t.detach().numpy()}
    names = []
    for name, p in clip.named_parameters():
        if p.grad is None:
            continue
        names.append(name)
        out["g_" + name] = cut(p.grad.numpy())
        out["n_" + name] = np.float64(p.grad.double().norm().item())
    out["names"] = np.array(names)
    save("mith_trunk_grads.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
