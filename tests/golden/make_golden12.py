#!/usr/bin/env python3
"""Twelfth golden generator — gradients of the REFERENCE's MITH HashingModel (model/MITH.py:399-453) by torch autograd on the
CPU: L = sum over the eight outputs of <output, fixed random cotangent>, differentiated w.r.t. the four inputs (token
matrices, cls / eos features) and every parameter.  Inputs / weights: tests/mithutil.py (make_golden3.py's first case)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import mithutil as mu  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402

SLICE = 211


def cut(a):
    a = np.asarray(a).reshape(-1)
    return a[::SLICE].copy() if a.size > 4096 else a.copy()


KEYS = ("img_cls_hash", "txt_cls_hash", "res_img_cls", "res_txt_cls", "img_tokens_hash", "txt_tokens_hash", "trans_tokens_i",
        "trans_tokens_t")


def cotangents(od, seed=17):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(od[k].shape, generator=g) for k in KEYS}


def gen():
    mm = ref_import("model.MITH")
    out = {}
    for (Nb, L, K) in [(3, 12, 16)]:
        tag = f"N{Nb}_L{L}_K{K}"
        hm = mm.HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=K, **mu.ARGS))
        shapes = {k: tuple(v.shape) for k, v in hm.state_dict().items()}
        st = mu.fill_state(shapes, 100 + K)
        hm.load_state_dict({k: (t(st[k]) if k in st else v) for k, v in hm.state_dict().items()})
        hm.train()
        c = mu.hash_inputs(Nb, L, K)
        ins = {k: t(c[k]).requires_grad_() for k in ("img_tokens", "txt_tokens", "img_cls", "txt_eos")}
        od = hm(ins["img_tokens"], ins["txt_tokens"], ins["img_cls"], ins["txt_eos"], t(c["kpm"]))
        G = cotangents(od)
        sum((od[k] * G[k]).sum() for k in KEYS).backward()
        for k, v in ins.items():
            out[f"{tag}_d_{k}"] = cut(v.grad.numpy())
            out[f"{tag}_n_{k}"] = np.float64(v.grad.double().norm().item())
        names = []
        for name, p in hm.named_parameters():
            if p.grad is None:
                continue
            names.append(name)
            out[f"{tag}_g_{name}"] = cut(p.grad.numpy())
            out[f"{tag}_n_{name}"] = np.float64(p.grad.double().norm().item())
        out[f"{tag}_names"] = np.array(names)
    save("mith_hash_grads.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()
